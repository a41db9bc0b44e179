"""MulticlassClassification head - BASELINE config 1 plumbing (resnet18, no neck, 10 classes, CPU).

Not on the hot path ("config 1 uses stock PyTorch only"): this is a plain ``torch.nn`` module so
that ``SihlModel(backbone, None, [head])`` can be exercised without a GPU.  Reference:
src/sihl/heads/multiclass_classification.py:11-69 (same constructor, ``convs`` layout and state_dict keys)."""
from typing import Dict, List, Tuple

from torch import Tensor, nn
from torch.nn.functional import cross_entropy


def _conv_relu_bn(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.ReLU(inplace=True), nn.BatchNorm2d(cout))


class MulticlassClassification(nn.Module):
    def __init__(self, in_channels: List[int], num_classes: int, num_channels: int = 256, num_layers: int = 1,
                 level: int = 5, label_smoothing: float = 0.0) -> None:
        assert num_classes > 0, num_classes
        assert len(in_channels) > level, (len(in_channels), level)
        assert num_channels > 0 and num_layers > 0, (num_channels, num_layers)
        super().__init__()
        self.num_classes, self.level, self.label_smoothing = num_classes, level, label_smoothing
        chans = [in_channels[level]] + [num_channels] * num_layers
        self.convs = nn.Sequential(
            nn.Sequential(*[_conv_relu_bn(chans[i], chans[i + 1]) for i in range(num_layers)]),
            nn.Conv2d(num_channels, num_classes, kernel_size=1),
            nn.AdaptiveAvgPool2d(1),
            nn.Flatten(),
        )
        self.output_shapes = {"scores": ("batch_size", num_classes), "classes": ("batch_size",)}

    def forward(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        return self.convs(inputs[self.level]).softmax(dim=1).max(dim=1)

    def training_step(self, inputs: List[Tensor], target: Tensor) -> Tuple[Tensor, Dict[str, float]]:
        logits = self.convs(inputs[self.level])
        return cross_entropy(logits, target.to(logits.device), label_smoothing=self.label_smoothing), {}
