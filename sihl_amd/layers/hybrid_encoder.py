"""HybridEncoder neck on the HIP path (reference src/sihl/layers/hybrid_encoder.py:14-134, the RT-DETR style neck the
reference's examples select by default).

Every convolution - 1x1 projections, the top-down 1x1 convs, the stride-2 3x3 convs of the bottom-up path and of the
extra levels, the 1x1 / 3x3 branches of the CSPRep fusions - is the fused matrix-core conv + BatchNorm (+ SiLU) block of
`convblocks.py`, working on NHWC tensors (channel concatenation is then a last-dim cat and the token layout of the
attention layer a plain reshape).  The single transformer encoder layer on the coarsest level (256 tokens at 512^2)
runs on PyTorch-ROCm (rocBLAS / SDPA), as do the parameter-free glue ops (nearest x2, cat, add, SiLU) and the
identity BatchNorm of the RepVGG blocks.
"""
from typing import List

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from sihl_amd import ops
from sihl_amd.layers.convblocks import Conv2dNormActivation


def sine_embedding_2d_grid(height: int, width: int, dim: int, temperature: float = 10000.0, device=None) -> Tensor:
    """(h, w, dim) sinusoidal position code, channels [sin y | cos y | sin x | cos x] (reference utils/__init__.py:95-138)."""
    if dim % 4:
        raise ValueError(f"Embedding dimension must be divisible by 4, got {dim}")
    quarter = dim // 4
    freq = torch.exp(torch.arange(quarter, dtype=torch.float32, device=device)
                     * -(torch.log(torch.tensor(float(temperature))).item() / (quarter - 1)))
    ys = (torch.arange(height, dtype=torch.float32, device=device)[:, None, None] * freq).expand(height, width, quarter)
    xs = (torch.arange(width, dtype=torch.float32, device=device)[None, :, None] * freq).expand(height, width, quarter)
    return torch.cat([ys.sin(), ys.cos(), xs.sin(), xs.cos()], dim=-1)


def _bn_nhwc(bn: nn.BatchNorm2d, x: Tensor) -> Tensor:
    return ops.nhwc(bn(ops.nchw_view(x)))


class RepVGGBlock(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.conv1 = Conv2dNormActivation(num_channels, num_channels, 3, activation_layer=None)
        self.conv2 = Conv2dNormActivation(num_channels, num_channels, 1, activation_layer=None)
        self.identity = nn.BatchNorm2d(num_channels)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return F.silu(self.conv1.forward_nhwc(x) + self.conv2.forward_nhwc(x) + _bn_nhwc(self.identity, x))

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class CSPRepLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_layers: int = 3):
        super().__init__()
        self.conv1 = Conv2dNormActivation(in_channels, out_channels, 1, activation_layer=nn.SiLU)
        self.conv2 = Conv2dNormActivation(in_channels, out_channels, 1, activation_layer=nn.SiLU)
        self.bottlenecks = nn.Sequential(*[RepVGGBlock(out_channels) for _ in range(num_layers)])

    def forward_nhwc(self, x1: Tensor, x2: Tensor) -> Tensor:
        x = torch.cat([x1, x2], dim=-1)
        y = self.conv1.forward_nhwc(x)
        for block in self.bottlenecks:
            y = block.forward_nhwc(y)
        return y + self.conv2.forward_nhwc(x)

    def forward(self, x1: Tensor, x2: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x1), ops.nhwc(x2)))


class HybridEncoder(nn.Module):
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
                                       activation="gelu", batch_first=True, norm_first=True),
            num_layers=1, enable_nested_tensor=False)
        silu = nn.SiLU
        self.up_convs, self.up_fusions = nn.ModuleList(), nn.ModuleList()
        for _ in range(self.top_in_level, bottom_level, -1):
            self.up_convs.append(Conv2dNormActivation(out_channels, out_channels, 1, activation_layer=silu))
            self.up_fusions.append(CSPRepLayer(out_channels * 2, out_channels))
        self.extra_downscalers = nn.ModuleList(
            Conv2dNormActivation(out_channels, out_channels, 3, stride=2, activation_layer=silu)
            for _ in range(top_level - len(in_channels) + 1))
        self.down_convs, self.down_fusions = nn.ModuleList(), nn.ModuleList()
        for _ in range(bottom_level, top_level):
            self.down_convs.append(Conv2dNormActivation(out_channels, out_channels, 3, stride=2, activation_layer=silu))
            self.down_fusions.append(CSPRepLayer(out_channels * 2, out_channels))

    def _attend(self, x: Tensor) -> Tensor:
        """x (B, h, w, C) -> x + pos + encoder(x + pos): in NHWC the token sequence is a reshape."""
        B, h, w, C = x.shape
        pos = sine_embedding_2d_grid(h, w, self.num_channels, device=x.device)
        tokens = (x.float() + pos).reshape(B, h * w, C)
        tokens = tokens + self.encoder(tokens)
        return tokens.reshape(B, h, w, C).to(x.dtype)

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        xs = [proj.forward_nhwc(ops.nhwc(inputs[self.bottom_level + i]))
              for i, proj in enumerate(self.input_projections)]
        xs[-1] = self._attend(xs[-1])
        inner = [xs[-1]]
        for i, (conv, fuse) in enumerate(zip(self.up_convs, self.up_fusions)):
            high = conv.forward_nhwc(inner[0])
            inner[0] = high
            up = ops.nhwc(F.interpolate(ops.nchw_view(high), scale_factor=2))  # nearest
            inner.insert(0, fuse.forward_nhwc(up, xs[len(xs) - 2 - i]))
        for down in self.extra_downscalers:
            inner.append(down.forward_nhwc(inner[-1]))
        outs = [inner[0]]
        for i, (conv, fuse) in enumerate(zip(self.down_convs, self.down_fusions)):
            outs.append(fuse.forward_nhwc(conv.forward_nhwc(outs[-1]), inner[i + 1]))
        return [*inputs[: self.bottom_level], *(ops.nchw_view(t) for t in outs), *inputs[self.top_level + 1:]]
