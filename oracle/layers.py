"""Oracle (CPU, pure torch) restatement of the reference's neck layers.

TEST INFRASTRUCTURE ONLY.  Module trees mirror the reference so that
``state_dict`` keys and shapes are identical (a reference checkpoint loads
key-for-key); the arithmetic is written out functionally here.

Reference files followed (relative to /root/reference/src/sihl):
  layers/convblocks.py:37-117   ConvNormAct / SequentialConvBlocks
  layers/pooling.py:7-30        BlurPool2d
  layers/scalers.py:16-56       AntialiasedDownscaler / Interpolate / SimpleUpscaler
  layers/bifpn.py:10-97         FastNormalizedFusion / BiFPNLayer / BiFPN
  layers/fpn.py:8-55            FPN
  layers/hybrid_encoder.py:14-134   HybridEncoder / RepVGGBlock / CSPRepLayer (+ utils/__init__.py:95-138 sine embedding)
torchvision 0.21 building blocks restated from their documented structure
(SURVEY.md App. B): ops.Conv2dNormActivation, ops.MLP.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

_ACTS = {
    "relu": lambda: nn.ReLU(inplace=True),
    "silu": lambda: nn.SiLU(inplace=True),
    "sigmoid": nn.Sigmoid,
    "softplus": nn.Softplus,
    "softmax": lambda: nn.Softmax(dim=1),
}


class ConvNormAct(nn.Sequential):
    """conv -> activation -> norm, in that order (convblocks.py:53-85).

    Child indices: "0" conv, then the activation (if any), then the norm
    (if any).  The conv has a bias only when there is no norm unless ``bias``
    says otherwise (convblocks.py:62).  ``padding=0`` falls through to the
    "same" padding because of the ``or`` (convblocks.py:59).
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1,
                 groups=1, padding=None, norm="batch", act="relu", bias=None):
        pad = padding or ((kernel_size - 1) // 2 * dilation)
        use_bias = (norm is None) if bias is None else bool(bias)
        mods: List[nn.Module] = [
            nn.Conv2d(in_channels, out_channels, kernel_size, stride, pad,
                      dilation=dilation, groups=groups, bias=use_bias)
        ]
        if act is not None:
            mods.append(_ACTS[act]())
        if norm == "batch":
            mods.append(nn.BatchNorm2d(out_channels))
        elif norm == "group":
            mods.append(nn.GroupNorm(in_channels // 8, out_channels))
        super().__init__(*mods)


class SequentialConvBlocks(nn.Sequential):
    """num_layers ConvNormAct blocks, Identity when num_layers <= 0 (convblocks.py:96-117)."""

    def __init__(self, in_channels, out_channels, num_layers, kernel_size=3, **kw):
        if num_layers <= 0:
            super().__init__(nn.Identity())
            return
        chans = [in_channels] + [out_channels] * num_layers
        super().__init__(*[
            ConvNormAct(chans[i], chans[i + 1], kernel_size=kernel_size, **kw)
            for i in range(num_layers)
        ])


def binomial_taps(kernel_size: int) -> Tensor:
    """Coefficients of (x/2 + 1/2)^(k-1): [1,2,1]/4 for k=3 (pooling.py:16-19)."""
    taps = np.ones(1)
    for _ in range(kernel_size - 1):
        taps = np.convolve(taps, [0.5, 0.5])
    return torch.tensor(taps.astype(np.float32))


class BlurPool2d(nn.Module):
    """Reflect-pad then depthwise binomial filter with stride (pooling.py:7-26)."""

    def __init__(self, in_channels: int, kernel_size: int = 3, stride: int = 1):
        super().__init__()
        self.in_channels, self.kernel_size, self.stride = in_channels, kernel_size, stride
        self.pad = ((stride - 1) + (kernel_size - 1)) // 2  # pooling.py:29-30, dilation 1
        t = binomial_taps(kernel_size)
        k2d = torch.outer(t, t)[None, None]
        self.register_buffer("kernel", k2d.repeat(in_channels, 1, 1, 1))

    def forward(self, x: Tensor) -> Tensor:
        p = self.pad
        x = F.pad(x, [p, p, p, p], mode="reflect")
        return F.conv2d(x, self.kernel.to(x.dtype), stride=self.stride, groups=self.in_channels)


class AntialiasedDownscaler(nn.Sequential):
    """ConvNormAct then stride-2 blur (scalers.py:16-23)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, **kw):
        super().__init__(ConvNormAct(in_channels, out_channels, kernel_size, **kw),
                         BlurPool2d(out_channels, stride=2))


class Interpolate(nn.Module):
    """F.interpolate wrapper, bilinear by default, align_corners=False (scalers.py:36-47)."""

    def __init__(self, scale=None, size=None, mode="bilinear"):
        super().__init__()
        self.scale, self.size, self.mode = scale, size, mode

    def forward(self, x: Tensor) -> Tensor:
        return F.interpolate(x, scale_factor=self.scale, size=self.size, mode=self.mode)


class SimpleUpscaler(nn.Sequential):
    """bilinear x2 then ConvNormAct (scalers.py:50-56)."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__(Interpolate(scale=2), ConvNormAct(in_channels, out_channels, kernel_size))


class FastNormalizedFusion(nn.Module):
    """sum_i softmax(w)_i * x_i; weights init to ones (bifpn.py:10-17)."""

    def __init__(self, num_inputs: int = 2):
        super().__init__()
        self.weights = nn.Parameter(torch.ones(num_inputs))

    def forward(self, inputs: Sequence[Tensor]) -> Tensor:
        w = torch.softmax(self.weights, dim=0)
        out = w[0] * inputs[0]
        for i in range(1, len(inputs)):
            out = out + w[i] * inputs[i]
        return out


class BiFPNLayer(nn.Module):
    """One bidirectional pass over the level pyramid (bifpn.py:20-53)."""

    def __init__(self, out_channels: int, num_levels: int, **kw):
        super().__init__()
        assert num_levels > 1
        self.num_levels = num_levels
        n = num_levels - 1
        self.upscalers = nn.ModuleList(Interpolate(scale=2) for _ in range(n))
        self.up_fusions = nn.ModuleList(FastNormalizedFusion(2) for _ in range(n))
        self.up_convs = nn.ModuleList(ConvNormAct(out_channels, out_channels, **kw) for _ in range(n))
        self.downscalers = nn.ModuleList(
            AntialiasedDownscaler(out_channels, out_channels, **kw) for _ in range(n))
        self.down_fusions = nn.ModuleList(FastNormalizedFusion(3) for _ in range(n))
        self.down_convs = nn.ModuleList(ConvNormAct(out_channels, out_channels, **kw) for _ in range(n))

    def forward(self, feats: List[Tensor]) -> List[Tensor]:
        L = self.num_levels
        assert len(feats) == L
        # top-down: td[top] is the input itself; module k serves level L-2-k (bifpn.py:41-45)
        td: List[Optional[Tensor]] = [None] * L
        td[L - 1] = feats[L - 1]
        for k in range(L - 1):
            lvl = L - 2 - k
            up = self.upscalers[k](td[lvl + 1])
            td[lvl] = self.up_convs[k](self.up_fusions[k]([up, feats[lvl]]))
        # bottom-up: bu[0] = td[0]; module k produces level k+1 (bifpn.py:47-52)
        bu = [td[0]]
        for k in range(L - 1):
            down = self.downscalers[k](bu[k])
            bu.append(self.down_convs[k](self.down_fusions[k]([down, feats[k + 1], td[k + 1]])))
        return bu


class BiFPN(nn.Module):
    """Laterals + extra downscaled levels + num_layers BiFPNLayers (bifpn.py:56-97)."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int,
                 top_level: int, num_layers: int = 3, **kw):
        super().__init__()
        assert num_layers > 0 and 0 < bottom_level < top_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.out_channels = list(in_channels[:bottom_level]) + [out_channels] * (top_level - bottom_level + 1)
        self.lateral_connections = nn.ModuleList(
            ConvNormAct(c, out_channels, kernel_size=1, **kw)
            for c in in_channels[bottom_level: top_level + 1])
        self.downscalers = nn.ModuleList(
            AntialiasedDownscaler(out_channels, out_channels, **kw)
            for _ in range(top_level + 1 - len(in_channels)))
        self.layers = nn.Sequential(*[
            BiFPNLayer(out_channels, top_level - bottom_level + 1, **kw) for _ in range(num_layers)])

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        feats = [lat(inputs[self.bottom_level + i]) for i, lat in enumerate(self.lateral_connections)]
        for ds in self.downscalers:
            feats.append(ds(feats[-1]))
        outs = self.layers(feats)
        return list(inputs[: self.bottom_level]) + list(outs) + list(inputs[self.top_level + 1:])


class Conv2dNormActivation(nn.Sequential):
    """torchvision.ops.Conv2dNormActivation: conv(bias=False) -> norm -> act."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None,
                 groups=1, norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, dilation=1,
                 inplace=True, bias=None):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        mods: List[nn.Module] = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                           dilation=dilation, groups=groups, bias=bias)]
        if norm_layer is not None:
            mods.append(norm_layer(out_channels))
        if activation_layer is not None:
            mods.append(activation_layer(inplace=inplace) if inplace is not None else activation_layer())
        super().__init__(*mods)
        self.out_channels = out_channels


class MLP(nn.Sequential):
    """torchvision.ops.MLP: (Linear, norm, act, Dropout)* + Linear + Dropout."""

    def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                 inplace=None, bias=True, dropout=0.0):
        kw = {} if inplace is None else {"inplace": inplace}
        mods: List[nn.Module] = []
        d = in_channels
        for h in hidden_channels[:-1]:
            mods.append(nn.Linear(d, h, bias=bias))
            if norm_layer is not None:
                mods.append(norm_layer(h))
            mods.append(activation_layer(**kw))
            mods.append(nn.Dropout(dropout, **kw))
            d = h
        mods.append(nn.Linear(d, hidden_channels[-1], bias=bias))
        mods.append(nn.Dropout(dropout, **kw))
        super().__init__(*mods)


class FPN(nn.Module):
    """Feature pyramid network (fpn.py:8-55).

    Quirk kept from fpn.py:43-48: the 1x1 ``up_conv`` REPLACES the map at its
    own level before that map is both upsampled (nearest x2) and later fed to
    its 3x3 output conv.
    """

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int):
        super().__init__()
        assert 0 < bottom_level < top_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.in_levels = range(bottom_level, min(top_level + 1, len(in_channels)))
        n_out = top_level - bottom_level + 1
        self.out_channels = list(in_channels)
        self.out_channels[bottom_level: top_level + 1] = [out_channels] * n_out
        C = Conv2dNormActivation
        self.input_projections = nn.ModuleList(C(in_channels[l], out_channels, 1) for l in self.in_levels)
        self.up_convs = nn.ModuleList(C(out_channels, out_channels, 1) for _ in self.in_levels[:-1])
        self.extra_downscalers = nn.ModuleList(
            C(out_channels, out_channels, stride=2) for _ in range(top_level - len(in_channels) + 1))
        self.out_convs = nn.ModuleList(C(out_channels, out_channels) for _ in range(n_out))

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        lo, hi = self.in_levels.start, self.in_levels.stop
        xs = [p(x) for p, x in zip(self.input_projections, inputs[lo:hi])]
        td = [xs[-1]]  # coarsest first
        for i, conv in enumerate(self.up_convs):
            td[i] = conv(td[i])
            td.append(F.interpolate(td[i], scale_factor=2) + xs[-(i + 2)])
        td = td[::-1]
        for down in self.extra_downscalers:
            td.append(down(td[-1]))
        outs = [conv(t) for conv, t in zip(self.out_convs, td)]
        return list(inputs[: self.bottom_level]) + outs + list(inputs[self.top_level + 1:])


# --------------------------------------------------------------------------- HybridEncoder (RT-DETR style neck)
def sine_embedding_2d_grid(height: int, width: int, dim: int, temperature: float = 10000.0, device=None) -> Tensor:
    """(h, w, dim) sinusoidal position code: first half of the channels encodes y, second half x, each half
    [sin | cos] over dim/4 geometric frequencies (utils/__init__.py:95-138)."""
    assert dim % 4 == 0
    quarter = dim // 4
    freq = torch.exp(torch.arange(quarter, dtype=torch.float32, device=device) * -(np.log(temperature) / (quarter - 1)))
    ys = torch.arange(height, dtype=torch.float32, device=device)[:, None, None] * freq
    xs = torch.arange(width, dtype=torch.float32, device=device)[None, :, None] * freq
    ys, xs = ys.expand(height, width, quarter), xs.expand(height, width, quarter)
    return torch.cat([ys.sin(), ys.cos(), xs.sin(), xs.cos()], dim=-1)


class RepVGGBlock(nn.Module):
    """silu(conv3x3+BN(x) + conv1x1+BN(x) + BN(x))  (hybrid_encoder.py:108-117)."""

    def __init__(self, num_channels: int):
        super().__init__()
        self.conv1 = Conv2dNormActivation(num_channels, num_channels, 3, activation_layer=None)
        self.conv2 = Conv2dNormActivation(num_channels, num_channels, 1, activation_layer=None)
        self.identity = nn.BatchNorm2d(num_channels)

    def forward(self, x: Tensor) -> Tensor:
        return F.silu(self.conv1(x) + self.conv2(x) + self.identity(x))


class CSPRepLayer(nn.Module):
    """Two 1x1 conv+BN+SiLU branches of the concatenated pair; one goes through RepVGG blocks (:120-134)."""

    def __init__(self, in_channels: int, out_channels: int, num_layers: int = 3):
        super().__init__()
        self.conv1 = Conv2dNormActivation(in_channels, out_channels, 1, activation_layer=nn.SiLU)
        self.conv2 = Conv2dNormActivation(in_channels, out_channels, 1, activation_layer=nn.SiLU)
        self.bottlenecks = nn.Sequential(*[RepVGGBlock(out_channels) for _ in range(num_layers)])

    def forward(self, x1: Tensor, x2: Tensor) -> Tensor:
        x = torch.cat([x1, x2], dim=1)
        return self.bottlenecks(self.conv1(x)) + self.conv2(x)


class HybridEncoder(nn.Module):
    """1x1 projections, one pre-norm transformer encoder layer on the coarsest projected level (with a sinusoidal
    position code, residual around the whole encoder), top-down path (1x1 conv, nearest x2, CSPRep fusion), optional
    stride-2 extra levels, bottom-up path (stride-2 3x3 conv, CSPRep fusion)  (hybrid_encoder.py:14-105)."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int):
        super().__init__()
        assert out_channels % 2 == 0
        self.in_channels = in_channels
        self.top_in_level = min(top_level, len(in_channels) - 1)
        self.bottom_level, self.top_level = bottom_level, top_level
        self.num_channels = out_channels
        self.out_channels = list(in_channels)
        self.out_channels[bottom_level: top_level + 1] = [out_channels] * (top_level - bottom_level + 1)
        self.input_projections = nn.ModuleList(
            Conv2dNormActivation(in_channels[l], out_channels, 1, activation_layer=None)
            for l in range(bottom_level, self.top_in_level + 1))
        self.encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(out_channels, nhead=8, dim_feedforward=4 * out_channels, dropout=0,
                                       activation="gelu", batch_first=True, norm_first=True), num_layers=1)
        self.up_convs, self.up_fusions = nn.ModuleList(), nn.ModuleList()
        for _ in range(self.top_in_level, bottom_level, -1):
            self.up_convs.append(Conv2dNormActivation(out_channels, out_channels, 1, activation_layer=nn.SiLU))
            self.up_fusions.append(CSPRepLayer(out_channels * 2, out_channels))
        self.extra_downscalers = nn.ModuleList(
            Conv2dNormActivation(out_channels, out_channels, 3, stride=2, activation_layer=nn.SiLU)
            for _ in range(top_level - len(in_channels) + 1))
        self.down_convs, self.down_fusions = nn.ModuleList(), nn.ModuleList()
        for _ in range(bottom_level, top_level):
            self.down_convs.append(Conv2dNormActivation(out_channels, out_channels, 3, stride=2, activation_layer=nn.SiLU))
            self.down_fusions.append(CSPRepLayer(out_channels * 2, out_channels))

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        xs = [proj(x) for proj, x in zip(self.input_projections, inputs[self.bottom_level: self.top_in_level + 1])]
        B, C, h, w = xs[-1].shape
        pos = sine_embedding_2d_grid(h, w, self.num_channels, device=xs[-1].device).permute(2, 0, 1)[None]
        tokens = (xs[-1] + pos).flatten(2).transpose(1, 2)
        tokens = tokens + self.encoder(tokens)
        xs[-1] = tokens.transpose(1, 2).reshape(B, C, h, w)
        inner = [xs[-1]]
        for i, (conv, fuse) in enumerate(zip(self.up_convs, self.up_fusions)):
            high = conv(inner[0])
            inner[0] = high
            inner.insert(0, fuse(F.interpolate(high, scale_factor=2), xs[len(xs) - 2 - i]))
        for down in self.extra_downscalers:
            inner.append(down(inner[-1]))
        outs = [inner[0]]
        for i, (conv, fuse) in enumerate(zip(self.down_convs, self.down_fusions)):
            outs.append(fuse(conv(outs[-1]), inner[i + 1]))
        return list(inputs[: self.bottom_level]) + outs + list(inputs[self.top_level + 1:])
