"""Level-list ResNet backbone (reference src/sihl/torchvision_backbone.py:42-49,102-210).

torchvision's published ResNet architecture (torchvision itself is absent here), honouring the reference's
level contract: level 0 is the input, level 1 = ``relu`` BEFORE max-pool, levels 2..5 = ``layer1..layer4``,
levels above 5 come from AntialiasedDownscaler blocks.  Parameter names follow torchvision's under ``model.``.

Execution: on a HIP device the residual stages (layer1..layer4 - every 1x1 / 3x3 / strided conv + BatchNorm + ReLU +
residual merge, forward and backward) run on the sihl HIP kernels (``native=None`` or ``True``): conv -> BN -> ReLU is
the same fused conv block the FPN uses, and the block tail is one normalise + add + ReLU pass - SURVEY §8(f) rank 2,
parity-tested against the oracle ResNet in tests/test_gpu_backbone.py.  ``native=False`` runs the trunk on
PyTorch-ROCm (MIOpen/CK), which SURVEY §8 a2 allows ("not a hand-kernel target").  Measured on the flagship step
(round 1, bs 32, 512^2, bf16): 689 img/s native vs 666 img/s MIOpen, and native has no ~45 s MIOpen JIT in the first
iteration.  The stem (conv1 7x7 / stride 2 over 3 channels -> bn1 -> relu -> max-pool): in bf16 (autocast) the native mode
runs all of it on the sihl kernels - the conv and its weight gradient on csrc/stem.hip (round 3, ops.StemFn: 130 + 115 us
against 254 + 244 us for MIOpen's kernels and a 60 us statistics pass); in fp32 the conv stays on PyTorch-ROCm and its
BatchNorm + ReLU and the max-pool run on the sihl kernels (ops.bn_act_train, ops.maxpool3x3s2).  CPU tensors (BASELINE
config 1, "stock PyTorch plumbing") always run through plain torch ops with the same parameters.
"""
import os
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.scalers import AntialiasedDownscaler


def _conv_bn(x: Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, act, training: bool, residual=None, hand_over=None,
             take_over=None, dx_to=None) -> Tensor:
    """conv -> BatchNorm -> act on NHWC tensors through the fused HIP conv block (torchvision order); with
    ``residual`` the block tail relu(BN(conv(x)) + residual) is one normalise+add+ReLU pass."""
    if training:
        ops.bump_counter(bn.num_batches_tracked)
    return ops.conv_block(x, conv.weight, None, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          stride=conv.stride[0], pad=conv.padding[0], dil=conv.dilation[0], act=act,
                          order="norm_act", training=training, eps=bn.eps, momentum=bn.momentum, residual=residual,
                          hand_over=hand_over, take_over=take_over, dx_to=dx_to)


def _carrier(block, x: Tensor, training: bool):
    """A GradCarrier for identity blocks in training: the shortcut's gradient then rides on conv1's dgrad.  Needs the
    first conv to produce an input gradient of x's shape (stride 1, x requires grad, conv1 trainable path)."""
    ok = (training and block.downsample is None and torch.is_grad_enabled() and x.requires_grad
          and block.conv1.stride[0] == 1 and not os.environ.get("SIHL_NO_GRAD_CARRIER"))  # env: A/B switch
    return ops.GradCarrier() if ok else None


def _proj_carrier(block, x: Tensor, training: bool):
    """A GradCarrier for projection blocks in training: the 1x1 (possibly strided) downsample conv parks its input
    gradient at its own output resolution and conv1's dgrad adds it at the pixels the projection reads - instead of a
    zero-dilated dgrad over the block input plus autograd's add of the two branch gradients."""
    ds = block.downsample[0]
    ok = (training and torch.is_grad_enabled() and x.requires_grad and block.conv1.stride[0] == 1
          and ds.kernel_size == (1, 1) and ds.padding == (0, 0) and not os.environ.get("SIHL_NO_GRAD_CARRIER"))
    return ops.GradCarrier() if ok else None


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

    def forward_nhwc(self, x):
        t = self.training
        idt = x if self.downsample is None else _conv_bn(x, self.downsample[0], self.downsample[1], None, t)
        c = _carrier(self, x, t)
        y = _conv_bn(x, self.conv1, self.bn1, "relu", t, take_over=c)
        return _conv_bn(y, self.conv2, self.bn2, None, t, residual=idt, hand_over=c)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
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

    def forward_nhwc(self, x):
        t = self.training
        if self.downsample is None:
            c = _carrier(self, x, t)
            y = _conv_bn(x, self.conv1, self.bn1, "relu", t, take_over=c)
            y = _conv_bn(y, self.conv2, self.bn2, "relu", t)
            return _conv_bn(y, self.conv3, self.bn3, None, t, residual=x, hand_over=c)
        c = _proj_carrier(self, x, t)
        y = _conv_bn(x, self.conv1, self.bn1, "relu", t, take_over=c)
        y = _conv_bn(y, self.conv2, self.bn2, "relu", t)
        # the projection runs AFTER conv1 so that autograd visits it first in backward (later nodes first): its compact
        # input gradient is parked before conv1's dgrad picks it up
        idt = _conv_bn(x, self.downsample[0], self.downsample[1], None, t, dx_to=c)
        return _conv_bn(y, self.conv3, self.bn3, None, t, residual=idt)


RESNETS = {"resnet18": (_Basic, [2, 2, 2, 2]), "resnet34": (_Basic, [3, 4, 6, 3]),
           "resnet50": (_Bottleneck, [3, 4, 6, 3]), "resnet101": (_Bottleneck, [3, 4, 23, 3]),
           "resnet152": (_Bottleneck, [3, 8, 36, 3])}


class _Trunk(nn.Module):
    def __init__(self, name: str, input_channels: int):
        super().__init__()
        block, depths = RESNETS[name]
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
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x: Tensor, n_taps: int, native: bool = False) -> List[Tensor]:
        if native and ops.stem_supported(x, self.conv1, self.bn1) \
                and (self.maxpool.kernel_size, self.maxpool.stride, self.maxpool.padding) == (3, 2, 1):
            # bf16: the whole stem on the sihl kernels (conv1 on csrc/stem.hip, statistics from its epilogue)
            stem = ops.stem_conv_bn_act(x, self.conv1, self.bn1, "relu")
            taps = [ops.nchw_view(stem)]
            y = ops.maxpool3x3s2(stem)
            return self._stages(y, taps, n_taps, native)
        z = self.conv1(x)
        vec = 8 if z.dtype == torch.bfloat16 else 4
        if native and self.bn1.training and torch.is_grad_enabled() and z.is_cuda and z.dtype in (torch.bfloat16, torch.float32) \
                and z.shape[1] % vec == 0 and self.bn1.momentum is not None and self.bn1.track_running_stats \
                and self.bn1.weight.requires_grad and not os.environ.get("SIHL_ATEN_STEM_BN"):  # env: A/B switch
            # fp32 (or an unsupported stem): the conv stays MIOpen's; its BatchNorm + ReLU, forward and backward, run on
            # the sihl kernels: statistics + finalize + one normalise pass instead of MIOpen's three BN kernels and an
            # ATen ReLU, and one reduce + one apply pass backward instead of threshold_backward + two BN kernels
            taps = [ops.nchw_view(ops.bn_act_train(ops.nhwc(z), self.bn1, "relu"))]
        else:
            taps = [self.relu(self.bn1(z))]
        if native and taps[0].dtype in (torch.bfloat16, torch.float32) and taps[0].shape[1] % vec == 0 \
                and (self.maxpool.kernel_size, self.maxpool.stride, self.maxpool.padding) == (3, 2, 1) \
                and not os.environ.get("SIHL_ATEN_MAXPOOL"):  # env: A/B switch
            y = ops.maxpool3x3s2(ops.nhwc(taps[0]))  # NHWC in (a view: the stem runs channels_last), NHWC out
        else:
            y = self.maxpool(taps[0])
            if native:
                y = ops.nhwc(y)
        return self._stages(y, taps, n_taps, native)

    def _stages(self, y: Tensor, taps: List[Tensor], n_taps: int, native: bool) -> List[Tensor]:
        for i in range(1, 5):
            if len(taps) >= n_taps:
                break
            if native:
                for block in getattr(self, f"layer{i}"):
                    y = block.forward_nhwc(y)
                taps.append(ops.nchw_view(y))
            else:
                y = getattr(self, f"layer{i}")(y)
                taps.append(y)
        return taps[:n_taps]


class ResNetBackbone(nn.Module):
    def __init__(self, name: str = "resnet50", pretrained: bool = False, input_channels: int = 3,
                 top_level: int = 5, frozen_levels: int = 0, native: Optional[bool] = None):
        """native: None (default) = sihl HIP kernels for the residual stages whenever the input is on a HIP device;
        True = require them (error on CPU input); False = PyTorch ops (MIOpen on ROCm)."""
        super().__init__()
        self.native = native
        if name not in RESNETS:
            raise ValueError(f"Architecture {name} is not supported. Select from {tuple(RESNETS)}")
        if pretrained:
            raise RuntimeError("pretrained weights need network access, which this environment does not have")
        self.name, self.top_level = name, top_level
        self.model = _Trunk(name, input_channels)
        self.n_taps = min(top_level, 5)
        min_size = 2 ** (top_level + 1)
        self.dummy_input = torch.zeros(1, input_channels, min_size, min_size)
        with torch.no_grad():
            was = self.model.training
            self.model.eval()
            self.out_channels = [input_channels] + [t.shape[1] for t in self.model(self.dummy_input, self.n_taps)]
            self.model.train(was)
        c = self.out_channels[-1]
        extra = range(top_level - 5)
        self.out_channels += [c for _ in extra]
        self.downscalers = nn.ModuleList([AntialiasedDownscaler(c, c) for _ in extra])
        self.freeze_levels(frozen_levels)

    def freeze_levels(self, num_levels: int) -> None:
        """Freeze the modules feeding the first ``num_levels`` levels (reference :189-210); <0 freezes all."""
        names = ["conv1", "bn1", "layer1", "layer2", "layer3", "layer4"]
        upto = {0: 0, 1: 2}.get(num_levels, min(num_levels + 1, len(names))) if num_levels >= 0 else len(names)
        for i, n in enumerate(names):
            for p in getattr(self.model, n).parameters():
                p.requires_grad_(i >= upto)

    def forward(self, input: Tensor) -> List[Tensor]:
        assert input.shape[2] % 2 ** self.top_level == 0
        assert input.shape[3] % 2 ** self.top_level == 0
        H, W = input.shape[2:]
        outs = [input]
        native = input.is_cuda if self.native is None else bool(self.native)
        if native and not input.is_cuda:
            raise RuntimeError("native=True needs a HIP device (no CPU fallback for the HIP residual stages)")
        if native and input.dim() == 4 and not input.is_contiguous(memory_format=torch.channels_last):
            input = input.contiguous(memory_format=torch.channels_last)
        for lvl, t in enumerate(self.model(input, self.n_taps, native), start=1):
            size = (H // 2 ** lvl, W // 2 ** lvl)
            outs.append(t if tuple(t.shape[2:]) == size else F.interpolate(t, size=size))
        for ds in self.downscalers:
            outs.append(ds(outs[-1]))
        return outs


TorchvisionBackbone = ResNetBackbone  # same constructor keywords as the reference class for resnets


# --------------------------------------------------------------------------------------------------------------------
# TimmBackbone level contract (reference src/sihl/timm_backbone.py:95-187; BASELINE configs[4]: timm convnext_base).
# timm is not available offline, so the trunk is this file's own statement of the PUBLISHED ConvNeXt architecture
# (stem 4x4 / stride 4 + LayerNorm; per stage: [LayerNorm + 2x2 / stride 2 conv between stages,] blocks of depthwise
# 7x7 -> LayerNorm -> Linear x4 -> GELU -> Linear -> layer scale -> residual), run on PyTorch-ROCm ops: backbones are
# third-party code in the reference and not a hand-kernel target.  Parity of the trunk against timm is
# UNPINNED; what IS the reference's own code - and is tested - is the level contract around it.
CONVNEXTS = {"convnext_tiny": ((3, 3, 9, 3), (96, 192, 384, 768)), "convnext_small": ((3, 3, 27, 3), (96, 192, 384, 768)),
             "convnext_base": ((3, 3, 27, 3), (128, 256, 512, 1024))}


class _LayerNorm2d(nn.LayerNorm):
    def forward(self, x: Tensor) -> Tensor:  # NCHW in / out, normalised over channels
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class _ConvNeXtBlock(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.fc1, self.act, self.fc2 = nn.Linear(dim, 4 * dim), nn.GELU(), nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(1e-6 * torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        y = self.conv_dw(x).permute(0, 2, 3, 1)
        y = self.fc2(self.act(self.fc1(self.norm(y)))) * self.gamma
        return x + y.permute(0, 3, 1, 2)


class _ConvNeXtTrunk(nn.Module):
    """features_only trunk: forward(x) -> the four stage outputs, at reductions 4, 8, 16, 32 (no stride-2 map)."""
    reductions = (4, 8, 16, 32)

    def __init__(self, depths, dims, input_channels: int = 3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(input_channels, dims[0], 4, stride=4), _LayerNorm2d(dims[0], eps=1e-6))
        stages = []
        for i, (d, c) in enumerate(zip(depths, dims)):
            down = nn.Identity() if i == 0 else nn.Sequential(_LayerNorm2d(dims[i - 1], eps=1e-6),
                                                             nn.Conv2d(dims[i - 1], c, 2, stride=2))
            stages.append(nn.Sequential(down, *[_ConvNeXtBlock(c) for _ in range(d)]))
        self.stages = nn.ModuleList(stages)

    def forward(self, x: Tensor) -> List[Tensor]:
        x = self.stem(x)
        outs = []
        for s in self.stages:
            x = s(x)
            outs.append(x)
        return outs


class TimmBackbone(nn.Module):
    """Level-list backbone with the reference TimmBackbone's contract (timm_backbone.py:95-187): ``out_channels`` =
    [input] + one entry per level; a trunk without a stride-2 feature map gets a FAKE level 1 - the (normalised) input
    nearest-resized to half size (:143-152) - so convnext_base yields [3, 3, 128, 256, 512, 1024]; every trunk output
    is nearest-resized to (H / 2^level, W / 2^level) (:178-184, a no-op for these trunks); levels above 5 come from
    AntialiasedDownscaler blocks (:167-171,185-186).  ``depths`` overrides the stage depths (tests)."""

    def __init__(self, name: str = "convnext_base", pretrained: bool = False, input_channels: int = 3,
                 top_level: int = 5, frozen_levels: int = 0, freeze_batchnorms: bool = False, depths=None):
        super().__init__()
        if name not in CONVNEXTS:
            raise ValueError(f"Architecture {name} is not supported. Select from {tuple(CONVNEXTS)}")
        if pretrained:
            raise RuntimeError("pretrained weights need network access, which this environment does not have")
        self.name, self.top_level = name, top_level
        d, dims = CONVNEXTS[name]
        self.model = _ConvNeXtTrunk(depths or d, dims, input_channels)
        self.normalize = nn.Identity()  # ImageNet normalisation only accompanies pretrained weights (:137-141)
        self.fake_level1 = 2 not in self.model.reductions
        min_size = 2 ** (top_level + 1)
        self.dummy_input = torch.zeros(1, input_channels, min_size, min_size)
        with torch.no_grad():
            self.out_channels = [input_channels] + [t.shape[1] for t in self._features(self.dummy_input)]
        c = self.out_channels[-1]
        extra = range(top_level - 5)
        self.out_channels += [c for _ in extra]
        self.downscalers = nn.ModuleList([AntialiasedDownscaler(c, c) for _ in extra])

    def _features(self, x: Tensor) -> List[Tensor]:
        feats = self.model(x)
        if self.fake_level1:
            feats = [F.interpolate(x, size=(x.shape[2] // 2, x.shape[3] // 2))] + feats
        return feats

    def forward(self, input: Tensor) -> List[Tensor]:
        assert input.shape[2] % 2 ** self.top_level == 0
        assert input.shape[3] % 2 ** self.top_level == 0
        x = self.normalize(input)
        H, W = x.shape[2:]
        outs = [input]
        for lvl, t in zip(range(1, self.top_level + 1), self._features(x)):
            size = (H // 2 ** lvl, W // 2 ** lvl)
            outs.append(t if tuple(t.shape[2:]) == size else F.interpolate(t, size=size))
        for ds in self.downscalers:
            outs.append(ds(outs[-1]))
        return outs
