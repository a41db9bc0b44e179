"""FPN neck on the HIP path (reference src/sihl/layers/fpn.py:8-55)."""
from typing import List

from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.convblocks import Conv2dNormActivation


class FPN(nn.Module):
    """https://arxiv.org/abs/1612.03144 - conv -> BN -> ReLU blocks (torchvision order)."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int):
        super().__init__()
        assert 0 < bottom_level < top_level
        self.in_levels = range(bottom_level, min(top_level + 1, len(in_channels)))
        self.bottom_level, self.top_level = bottom_level, top_level
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
        xs = [p.forward_nhwc(ops.nhwc(x)) for p, x in zip(self.input_projections, inputs[lo:hi])]
        td = [xs[-1]]
        for i, conv in enumerate(self.up_convs):  # fpn.py:43-48: the 1x1 REPLACES the map at its own level
            td[i] = conv.forward_nhwc(td[i])
            td.append(ops.nearest_up2_add(td[i], xs[-(i + 2)]))
        td = td[::-1]
        for down in self.extra_downscalers:
            td.append(down.forward_nhwc(td[-1]))
        outs = [ops.nchw_view(conv.forward_nhwc(t)) for conv, t in zip(self.out_convs, td)]
        return list(inputs[: self.bottom_level]) + outs + list(inputs[self.top_level + 1:])
