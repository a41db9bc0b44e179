"""Oracle (CPU, pure torch): SihlModel and the ResNet level-list backbone.

TEST INFRASTRUCTURE ONLY.  Reference files followed (relative to
/root/reference/src/sihl):
  sihl_model.py:6-25                  SihlModel
  torchvision_backbone.py:42-49       ResNet taps: relu, layer1..layer4
  torchvision_backbone.py:102-186     level contract (input prepended as level 0,
                                      nearest-resize to H/2^l, extra downscalers above 5)
torchvision itself is absent from the container; the ResNet below is a plain
torch.nn restatement of the published architecture ("parity unpinned").
"""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from oracle.layers import AntialiasedDownscaler


class SihlModel(nn.Module):
    def __init__(self, backbone: nn.Module, neck: Optional[nn.Module], heads: List[nn.Module]):
        super().__init__()
        self.backbone, self.neck, self.heads = backbone, neck, nn.ModuleList(heads)

    def extract_features(self, input: Tensor) -> List[Tensor]:
        x = self.backbone(input)
        return x if self.neck is None else self.neck(x)

    def forward(self, input: Tensor):
        x = self.extract_features(input)
        return [head(x) for head in self.heads]


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = None
        if stride != 1 or cin != width:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width, 1, stride, bias=False), nn.BatchNorm2d(width))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)  # stride on the 3x3 (v1.5)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


_RESNETS = {"resnet18": (_Basic, [2, 2, 2, 2]), "resnet34": (_Basic, [3, 4, 6, 3]),
            "resnet50": (_Bottleneck, [3, 4, 6, 3]), "resnet101": (_Bottleneck, [3, 4, 23, 3])}


class _ResNetTrunk(nn.Module):
    def __init__(self, name: str, input_channels: int):
        super().__init__()
        block, depths = _RESNETS[name]
        self.conv1 = nn.Conv2d(input_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (w, n) in enumerate(zip([64, 128, 256, 512], depths)):
            blocks = []
            for j in range(n):
                blocks.append(block(cin, w, (2 if i > 0 else 1) if j == 0 else 1))
                cin = w * block.expansion
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        for m in self.modules():  # torchvision's default init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x: Tensor, n_taps: int) -> List[Tensor]:
        taps = [self.relu(self.bn1(self.conv1(x)))]  # level 1 = relu BEFORE maxpool
        y = self.maxpool(taps[0])
        for i in range(1, 5):
            if len(taps) >= n_taps:
                break
            y = getattr(self, f"layer{i}")(y)
            taps.append(y)
        return taps[:n_taps]


class ResNetBackbone(nn.Module):
    """Level-list backbone: [input, relu, layer1, layer2, layer3, layer4, extra...]."""

    def __init__(self, name: str = "resnet50", pretrained: bool = False, input_channels: int = 3,
                 top_level: int = 5, frozen_levels: int = 0):
        super().__init__()
        if name not in _RESNETS:
            raise ValueError(f"Architecture {name} is not supported. Select from {tuple(_RESNETS)}")
        if pretrained:
            raise RuntimeError("pretrained weights need network access; unavailable offline")
        self.name, self.top_level = name, top_level
        self.model = _ResNetTrunk(name, input_channels)
        self.n_taps = min(top_level, 5)
        self.dummy_input = torch.zeros(1, input_channels, 2 ** (top_level + 1), 2 ** (top_level + 1))
        with torch.no_grad():
            was = self.model.training
            self.model.eval()
            self.out_channels = [input_channels] + [t.shape[1] for t in self.model(self.dummy_input, self.n_taps)]
            self.model.train(was)
        extra = range(top_level - 5)
        c = self.out_channels[-1]
        self.out_channels += [c for _ in extra]
        self.downscalers = nn.ModuleList([AntialiasedDownscaler(c, c) for _ in extra])

    def forward(self, input: Tensor) -> List[Tensor]:
        assert input.shape[2] % 2 ** self.top_level == 0 and input.shape[3] % 2 ** self.top_level == 0
        H, W = input.shape[2:]
        outs = [input]
        for lvl, t in enumerate(self.model(input, self.n_taps), start=1):
            size = (H // 2 ** lvl, W // 2 ** lvl)
            outs.append(t if tuple(t.shape[2:]) == size else F.interpolate(t, size=size))
        for ds in self.downscalers:
            outs.append(ds(outs[-1]))
        return outs


# --------------------------------------------------------------------------------------------------------------------
# TimmBackbone level contract (reference src/sihl/timm_backbone.py:95-187) around a ConvNeXt-shaped trunk restated from
# the published architecture (timm 1.0.15 is not in the image: the trunk is "parity unpinned", the contract - fake
# level 1, out_channels, nearest resize to the level size, extra downscalers - follows the reference's own lines).
_CONVNEXTS = {"convnext_tiny": ((3, 3, 9, 3), (96, 192, 384, 768)), "convnext_small": ((3, 3, 27, 3), (96, 192, 384, 768)),
              "convnext_base": ((3, 3, 27, 3), (128, 256, 512, 1024))}


class _LN2d(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class _CNBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.fc1, self.act, self.fc2 = nn.Linear(dim, 4 * dim), nn.GELU(), nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(1e-6 * torch.ones(dim))

    def forward(self, x):
        y = self.conv_dw(x).permute(0, 2, 3, 1)
        return x + (self.fc2(self.act(self.fc1(self.norm(y)))) * self.gamma).permute(0, 3, 1, 2)


class _CNTrunk(nn.Module):
    reductions = (4, 8, 16, 32)

    def __init__(self, depths, dims, input_channels=3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(input_channels, dims[0], 4, stride=4), _LN2d(dims[0], eps=1e-6))
        self.stages = nn.ModuleList(
            nn.Sequential(nn.Identity() if i == 0 else nn.Sequential(_LN2d(dims[i - 1], eps=1e-6),
                                                                     nn.Conv2d(dims[i - 1], c, 2, stride=2)),
                          *[_CNBlock(c) for _ in range(d)]) for i, (d, c) in enumerate(zip(depths, dims)))

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for s in self.stages:
            x = s(x)
            outs.append(x)
        return outs


class TimmBackbone(nn.Module):
    def __init__(self, name="convnext_base", pretrained=False, input_channels=3, top_level=5, frozen_levels=0,
                 freeze_batchnorms=False, depths=None):
        super().__init__()
        if name not in _CONVNEXTS:
            raise ValueError(f"Architecture {name} is not supported. Select from {tuple(_CONVNEXTS)}")
        if pretrained:
            raise RuntimeError("pretrained weights need network access; unavailable offline")
        self.name, self.top_level = name, top_level
        d, dims = _CONVNEXTS[name]
        self.model = _CNTrunk(depths or d, dims, input_channels)
        self.normalize = nn.Identity()
        self.fake_level1 = 2 not in self.model.reductions  # timm_backbone.py:143-152
        self.dummy_input = torch.zeros(1, input_channels, 2 ** (top_level + 1), 2 ** (top_level + 1))
        with torch.no_grad():
            self.out_channels = [input_channels] + [t.shape[1] for t in self._features(self.dummy_input)]
        c = self.out_channels[-1]
        extra = range(top_level - 5)
        self.out_channels += [c for _ in extra]
        self.downscalers = nn.ModuleList([AntialiasedDownscaler(c, c) for _ in extra])

    def _features(self, x):
        feats = self.model(x)
        if self.fake_level1:
            feats = [F.interpolate(x, size=(x.shape[2] // 2, x.shape[3] // 2))] + feats
        return feats

    def forward(self, input):
        assert input.shape[2] % 2 ** self.top_level == 0 and input.shape[3] % 2 ** self.top_level == 0
        x = self.normalize(input)
        H, W = x.shape[2:]
        outs = [input]
        for lvl, t in zip(range(1, self.top_level + 1), self._features(x)):
            size = (H // 2 ** lvl, W // 2 ** lvl)
            outs.append(t if tuple(t.shape[2:]) == size else F.interpolate(t, size=size))
        for ds in self.downscalers:
            outs.append(ds(outs[-1]))
        return outs
