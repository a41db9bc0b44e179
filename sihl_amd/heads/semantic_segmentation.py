"""SemanticSegmentation head on the HIP path (reference src/sihl/heads/semantic_segmentation.py:11-92,123-182;
PP-LiteSeg style: SPPM context -> per level [3x3 lateral, bilinear x2 + 3x3, UAFM fusion] -> conv tower + 1x1
classifier).  forward never materialises the (B, classes, H, W) tensor: nearest resize + softmax + max are one
kernel; training_step fuses the resize with the cross-entropy."""
from typing import Dict, List, Tuple, Union

from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.convblocks import ConvNormAct, SequentialConvBlocks, _ConvBlock
from sihl_amd.layers.scalers import Interpolate, SimpleUpscaler


class _PlainConv(_ConvBlock):
    """View of a bare nn.Conv2d (bias, no norm, no activation) through the shared conv-block forward."""

    def __init__(self, conv: nn.Conv2d):
        nn.Module.__init__(self)
        self._modules["0"] = conv


class SPPM(nn.Module):
    """https://arxiv.org/abs/2204.02681 - bilinear "pooling" pyramid (reference :123-160)."""

    def __init__(self, in_channels: int, out_channels: int, pool_sizes: Tuple[int] = (1, 2, 4),
                 with_shortcut: bool = False) -> None:
        super().__init__()
        self.with_shortcut = with_shortcut
        if with_shortcut:
            raise NotImplementedError("SPPM(with_shortcut=True) is not used by the reference head; not built")
        if len(pool_sizes) == 0:
            raise NotImplementedError("SPPM without pooling branches is outside the HIP hot path")
        self.pools = nn.ModuleList([nn.Sequential(Interpolate(size=p), ConvNormAct(in_channels, out_channels, 1))
                                    for p in pool_sizes])
        self.out_conv = ConvNormAct(out_channels, out_channels, 1)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        size = tuple(x.shape[1:3])
        acc = None
        for pool in self.pools:
            y = pool[1].forward_nhwc(pool[0].forward_nhwc(x))
            acc = ops.resize_bilinear(y, size, add=acc)
        return self.out_conv.forward_nhwc(acc)

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class UAFM(nn.Module):
    """https://arxiv.org/abs/2204.02681 - unified attention fusion (reference :163-182)."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.conv = ConvNormAct(4, 1, norm=None, act="sigmoid")  # parameter container: conv.0.{weight,bias}

    def forward_nhwc(self, x1: Tensor, x2: Tensor) -> Tensor:
        return ops.uafm(x1, x2, self.conv[0].weight, self.conv[0].bias)

    def forward(self, x1: Tensor, x2: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x1), ops.nhwc(x2)))


class SemanticSegmentation(nn.Module):
    """Semantic segmentation is pixelwise multiclass classification (PP-LiteSeg decoder)."""

    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3, top_level: int = 5,
                 num_channels: int = 256, num_layers: int = 3, pool_sizes: List[int] = [1, 2, 4],
                 ignore_index: Union[int, None] = None) -> None:
        assert num_classes > 0
        assert len(in_channels) > top_level >= bottom_level > 0
        assert num_channels > 0 and num_layers >= 0
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.num_channels, self.num_layers = num_channels, num_layers
        self.pool_sizes = tuple(pool_sizes)
        self.bottom_level, self.top_level = bottom_level, top_level
        self.ignore_index = ignore_index or -100  # reference :51 (index 0 cannot be ignored)
        self.levels = list(range(bottom_level, top_level + 1))
        self.rev_levels = list(reversed(range(bottom_level, top_level)))
        self.context_aggregation = SPPM(in_channels[top_level], num_channels, self.pool_sizes)
        self.lateral_convs = nn.ModuleList([ConvNormAct(in_channels[l], num_channels) for l in self.rev_levels])
        self.upscalers = nn.ModuleList([SimpleUpscaler(num_channels, num_channels) for _ in self.rev_levels])
        self.fusions = nn.ModuleList([UAFM(num_channels, num_channels) for _ in self.rev_levels])
        self.out_conv = nn.Sequential(SequentialConvBlocks(num_channels, num_channels, num_layers),
                                      nn.Conv2d(num_channels, num_classes, kernel_size=1))
        self.output_shapes = {"score_maps": ("batch_size", "height", "width"),
                              "class_maps": ("batch_size", "height", "width")}

    def _logits_nhwc(self, inputs: List[Tensor]) -> Tensor:
        x = self.context_aggregation.forward_nhwc(ops.nhwc(inputs[self.top_level]))
        for l, lat, up, fuse in zip(self.rev_levels, self.lateral_convs, self.upscalers, self.fusions):
            x = fuse.forward_nhwc(lat.forward_nhwc(ops.nhwc(inputs[l])), up.forward_nhwc(x))
        x = self.out_conv[0].forward_nhwc(x)
        return _PlainConv(self.out_conv[1]).forward_nhwc(x)

    def get_logits(self, inputs: List[Tensor]) -> Tensor:
        return ops.nchw_view(self._logits_nhwc(inputs))

    def forward(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        return ops.softmax_max_resize(self._logits_nhwc(inputs), tuple(inputs[0].shape[2:]))

    def training_step(self, inputs: List[Tensor], targets: Tensor) -> Tuple[Tensor, Dict[str, float]]:
        logits = self._logits_nhwc(inputs)
        return ops.ce_resize(logits, targets.to(logits.device), self.ignore_index), {}

    def on_validation_start(self) -> None:
        from sihl_amd.metrics import SegmentationConfusion

        self._val_losses: List[Tensor] = []
        self._val_confusion = SegmentationConfusion(self.num_classes, self.ignore_index)

    def validation_step(self, inputs: List[Tensor], targets: Tensor) -> Tuple[Tensor, Dict[str, float]]:
        """Loss + the reference's two metrics (:106-120): class maps at the targets' resolution from the fused nearest-resize
        + softmax-max kernel (the argmax of the scores is the argmax of the logits), confusion counts kept on the device."""
        import torch

        with torch.no_grad():
            logits = self._logits_nhwc(inputs)
            targets = targets.to(logits.device)
            loss = ops.ce_resize(logits, targets, self.ignore_index)
            _, classes = ops.softmax_max_resize(logits, tuple(targets.shape[1:]))
            self._val_confusion.update(classes, targets)
        self._val_losses.append(loss.detach())
        return loss, {}

    def on_validation_end(self) -> Dict[str, float]:
        import torch

        out = {"loss": torch.stack(self._val_losses).mean().item() if self._val_losses else float("nan")}
        out.update(self._val_confusion.compute())
        return out
