"""BiFPN neck on the HIP path (reference src/sihl/layers/bifpn.py:10-97).

Each fusion node runs as ONE kernel together with its producer: [bilinear x2 upsample (+) skip] for
the top-down path, [reflect blur-pool stride 2 (+) input (+) top-down] for the bottom-up path, so
the upsampled / blurred / stacked intermediates of the reference are never written to HBM.
"""
from typing import List, Optional

import torch
from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.convblocks import ConvNormAct
from sihl_amd.layers.scalers import AntialiasedDownscaler, Interpolate


class FastNormalizedFusion(nn.Module):
    def __init__(self, num_inputs: int = 2):
        super().__init__()
        if num_inputs not in (2, 3):
            raise NotImplementedError("2- and 3-input fusion nodes only")
        self.weights = nn.Parameter(torch.ones(num_inputs), requires_grad=True)

    def forward(self, inputs: List[Tensor]) -> Tensor:
        return ops.nchw_view(ops.fuse_sum(self.weights, [ops.nhwc(x) for x in inputs]))


class BiFPNLayer(nn.Module):
    def __init__(self, out_channels: int, num_levels: int, **kw):
        super().__init__()
        assert num_levels > 1, num_levels
        self.num_levels = num_levels
        n = num_levels - 1
        self.upscalers = nn.ModuleList(Interpolate(scale=2) for _ in range(n))
        self.up_fusions = nn.ModuleList(FastNormalizedFusion(2) for _ in range(n))
        self.up_convs = nn.ModuleList(ConvNormAct(out_channels, out_channels, **kw) for _ in range(n))
        self.downscalers = nn.ModuleList(AntialiasedDownscaler(out_channels, out_channels, **kw) for _ in range(n))
        self.down_fusions = nn.ModuleList(FastNormalizedFusion(3) for _ in range(n))
        self.down_convs = nn.ModuleList(ConvNormAct(out_channels, out_channels, **kw) for _ in range(n))

    def forward_nhwc(self, feats: List[Tensor]) -> List[Tensor]:
        L = self.num_levels
        assert len(feats) == L
        td: List[Optional[Tensor]] = [None] * L
        td[L - 1] = feats[L - 1]
        for k in range(L - 1):  # module k serves level L-2-k (bifpn.py:41-45)
            lvl = L - 2 - k
            fused = ops.fuse_up2(td[lvl + 1], feats[lvl], self.up_fusions[k].weights)
            td[lvl] = self.up_convs[k].forward_nhwc(fused)
        bu = [td[0]]
        for k in range(L - 1):  # module k produces level k+1 (bifpn.py:47-52)
            pre_blur = self.downscalers[k][0].forward_nhwc(bu[k])
            fused = ops.blur_fuse(pre_blur, feats[k + 1], td[k + 1], self.down_fusions[k].weights)
            bu.append(self.down_convs[k].forward_nhwc(fused))
        return bu

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        return [ops.nchw_view(t) for t in self.forward_nhwc([ops.nhwc(t) for t in inputs])]


class BiFPN(nn.Module):
    """https://arxiv.org/abs/1911.09070 - same constructor and level-list contract as the reference."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int,
                 num_layers: int = 3, **kw):
        super().__init__()
        assert num_layers > 0
        assert 0 < bottom_level < top_level
        self.out_channels = list(in_channels[:bottom_level]) + [out_channels] * (top_level - bottom_level + 1)
        self.bottom_level, self.top_level = bottom_level, top_level
        self.lateral_connections = nn.ModuleList(
            ConvNormAct(c, out_channels, kernel_size=1, **kw) for c in in_channels[bottom_level: top_level + 1])
        self.downscalers = nn.ModuleList(
            AntialiasedDownscaler(out_channels, out_channels, **kw) for _ in range(top_level + 1 - len(in_channels)))
        self.layers = nn.Sequential(*(BiFPNLayer(out_channels, top_level - bottom_level + 1, **kw)
                                      for _ in range(num_layers)))

    def forward(self, inputs: List[Tensor]) -> List[Tensor]:
        feats = [lat.forward_nhwc(ops.nhwc(inputs[self.bottom_level + i]))
                 for i, lat in enumerate(self.lateral_connections)]
        for ds in self.downscalers:
            feats.append(ds.forward_nhwc(feats[-1]))
        for layer in self.layers:
            feats = layer.forward_nhwc(feats)
        outs = [ops.nchw_view(t) for t in feats]
        return list(inputs[: self.bottom_level]) + outs + list(inputs[self.top_level + 1:])
