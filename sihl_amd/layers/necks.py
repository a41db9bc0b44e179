"""Feature-pyramid necks on the HIP path: BiFPN and FPN.

A neck maps a level list (index = level, level i has stride 2**i, level 0 = the image) to a level list of the
same length: levels ``bottom_level..top_level`` are replaced by ``out_channels``-wide maps, the others pass
through untouched; ``.out_channels`` lists the channel count per level.  Constructor arguments, attribute names
and therefore state_dict keys follow the reference (BiFPN: src/sihl/layers/bifpn.py:10-97, FPN:
src/sihl/layers/fpn.py:8-55) so that its checkpoints load key-for-key.

What differs is execution.  All work happens on NHWC tensors.  In BiFPN each fusion node is ONE kernel together
with its producer - [bilinear x2 upsample (+) skip] on the way down, [reflect blur-pool stride 2 (+) input (+)
top-down map] on the way up - so the upsampled / blurred / stacked intermediates the reference materialises never
reach HBM; the 3x3 convolutions are the matrix-core conv blocks of ``convblocks.py``.
"""
from typing import List, Sequence

import torch
from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.convblocks import Conv2dNormActivation, ConvNormAct
from sihl_amd.layers.scalers import AntialiasedDownscaler, Interpolate


def _splice(inputs: Sequence[Tensor], lo: int, hi: int, new_nhwc: Sequence[Tensor]) -> List[Tensor]:
    """inputs with levels lo..hi replaced by the (NHWC) tensors in new_nhwc, handed back NCHW-logical."""
    return [*inputs[:lo], *(ops.nchw_view(t) for t in new_nhwc), *inputs[hi + 1:]]


def _modules(n: int, make) -> nn.ModuleList:
    return nn.ModuleList([make(i) for i in range(n)])


# --------------------------------------------------------------------------------------------------- BiFPN
class FastNormalizedFusion(nn.Module):
    """sum_i softmax(weights)_i * x_i (the reference uses softmax, not the paper's ReLU normalisation)."""

    def __init__(self, num_inputs: int = 2) -> None:
        super().__init__()
        if num_inputs not in (2, 3):
            raise NotImplementedError("2- and 3-input fusion nodes only")
        self.weights = nn.Parameter(torch.ones(num_inputs))

    def forward(self, inputs: Sequence[Tensor]) -> Tensor:
        return ops.nchw_view(ops.fuse_sum(self.weights, [ops.nhwc(t) for t in inputs]))


class BiFPNLayer(nn.Module):
    """One top-down + bottom-up sweep over ``num_levels`` maps of equal width."""

    def __init__(self, out_channels: int, num_levels: int, **conv_kw) -> None:
        super().__init__()
        if num_levels < 2:
            raise AssertionError(num_levels)
        self.num_levels = num_levels
        hops = num_levels - 1  # one module of every kind per pair of neighbouring levels
        conv = lambda _: ConvNormAct(out_channels, out_channels, **conv_kw)  # noqa: E731
        self.upscalers = _modules(hops, lambda _: Interpolate(scale=2))  # parameter-free, kept for layout parity
        self.up_fusions = _modules(hops, lambda _: FastNormalizedFusion(2))
        self.up_convs = _modules(hops, conv)
        self.downscalers = _modules(hops, lambda _: AntialiasedDownscaler(out_channels, out_channels, **conv_kw))
        self.down_fusions = _modules(hops, lambda _: FastNormalizedFusion(3))
        self.down_convs = _modules(hops, conv)

    def forward_nhwc(self, feats: Sequence[Tensor], first_merged: Tensor = None, next_up_weights: Tensor = None,
                     return_next: bool = False):
        """first_merged: the first top-down node (fuse_up2 of the two top maps), when the caller already has it (the previous
        layer's last conv emitted it); next_up_weights: the NEXT layer's first top-down fusion weights - with return_next the
        result is (maps, that layer's first node or None).

        Inference on the small maps (conv_pyr.hip's shapes) runs each conv block together with the fusion node that consumes
        its output - the node is computed in the conv's epilogue from the workgroup's own channel slice of the whole map
        (ConvNormAct.forward_nhwc_emit) - so the top of the pyramid is one launch per conv; everywhere else, and in training,
        every node is its own kernel (or, opt-in, part of the consuming conv's loader: forward_fused_node)."""
        top = self.num_levels - 1
        if len(feats) != self.num_levels:
            raise AssertionError((len(feats), self.num_levels))
        # descend: module k works at level top-1-k and consumes the map produced one level above it
        down = {top: feats[top]}
        merged = first_merged
        for k in range(top):
            lvl = top - 1 - k
            conv, wts = self.up_convs[k], self.up_fusions[k].weights
            nxt = None  # the node one level further down, if this conv can emit it
            if merged is None:
                # node + conv block as one launch on the small maps (opt-in), else fusion kernel, then conv
                y = conv.forward_fused_node(("up2", down[lvl + 1], feats[lvl], wts))
                if y is not None:
                    down[lvl] = y
                    continue
                merged = ops.fuse_up2(down[lvl + 1], feats[lvl], wts)
            if k + 1 < top:
                got = conv.forward_nhwc_emit(merged, ("up2", feats[lvl - 1], self.up_fusions[k + 1].weights))
                if got is not None:
                    down[lvl], nxt = got
            if nxt is None:
                down[lvl] = conv.forward_nhwc(merged)
            merged = nxt
        # ascend: module k produces level k+1 from (blurred conv of level k, the input, the top-down map)
        out = [down[0]]
        next_first = None
        for k in range(top):
            w3 = self.down_fusions[k].weights
            merged = None
            # the downscaler's conv block with the bottom-up node in its epilogue (its own output has no other reader) ...
            got = self.downscalers[k][0].forward_nhwc_emit(out[k], ("blur", feats[k + 1], down[k + 1], w3), write_y=False)
            if got is not None:
                merged = got[1]
            else:
                # ... or the conv block; its blur - and in training its BatchNorm affine - is fused into the merge below
                affine = ops.DeferredAffine()
                pre = self.downscalers[k][0].forward_nhwc(out[k], defer=affine)
                y = self.down_convs[k].forward_fused_node(("blur", pre, feats[k + 1], down[k + 1], w3, affine))
                if y is not None:
                    out.append(y)
                    continue
                merged = ops.blur_fuse(pre, feats[k + 1], down[k + 1], w3, a_affine=affine)
            y = None
            if k == top - 1 and next_up_weights is not None:
                # the top map's conv also emits the NEXT layer's first top-down node: up2(new top) (+) new map below it
                got = self.down_convs[k].forward_nhwc_emit(merged, ("up2", out[k], next_up_weights))
                if got is not None:
                    y, next_first = got
            out.append(y if y is not None else self.down_convs[k].forward_nhwc(merged))
        return (out, next_first) if return_next else out

    def forward(self, inputs: Sequence[Tensor]) -> List[Tensor]:
        return [ops.nchw_view(t) for t in self.forward_nhwc([ops.nhwc(t) for t in inputs])]


class BiFPN(nn.Module):
    """EfficientDet's bidirectional pyramid (arXiv:1911.09070): 1x1 laterals, antialiased extra levels up to
    ``top_level``, then ``num_layers`` BiFPNLayers."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int,
                 num_layers: int = 3, **conv_kw) -> None:
        super().__init__()
        if not (num_layers > 0 and 0 < bottom_level < top_level):
            raise AssertionError((num_layers, bottom_level, top_level))
        self.bottom_level, self.top_level = bottom_level, top_level
        width = top_level - bottom_level + 1
        self.out_channels = [*in_channels[:bottom_level], *([out_channels] * width)]
        fed = in_channels[bottom_level: top_level + 1]  # levels the backbone provides
        self.lateral_connections = _modules(len(fed), lambda i: ConvNormAct(fed[i], out_channels, kernel_size=1, **conv_kw))
        self.downscalers = _modules(top_level + 1 - len(in_channels),
                                    lambda _: AntialiasedDownscaler(out_channels, out_channels, **conv_kw))
        self.layers = nn.Sequential(*[BiFPNLayer(out_channels, width, **conv_kw) for _ in range(num_layers)])

    def forward(self, inputs: Sequence[Tensor]) -> List[Tensor]:
        maps = [lat.forward_nhwc(ops.nhwc(inputs[self.bottom_level + i]))
                for i, lat in enumerate(self.lateral_connections)]
        for extra in self.downscalers:
            maps.append(extra.forward_nhwc(maps[-1]))
        first = None
        for i, layer in enumerate(self.layers):
            nxt = self.layers[i + 1].up_fusions[0].weights if i + 1 < len(self.layers) else None
            maps, first = layer.forward_nhwc(maps, first_merged=first, next_up_weights=nxt, return_next=True)
        return _splice(inputs, self.bottom_level, self.top_level, maps)


# ----------------------------------------------------------------------------------------------------- FPN
class FPN(nn.Module):
    """Lin et al.'s pyramid (arXiv:1612.03144) with torchvision-style conv -> BN -> ReLU blocks: 1x1 projections,
    a top-down path whose 1x1 ``up_convs`` REPLACE the map at their own level before it is upsampled (nearest x2)
    and added to the level below (a quirk of the reference kept on purpose), optional stride-2 extra levels, and a
    3x3 output conv per level."""

    def __init__(self, in_channels: List[int], out_channels: int, bottom_level: int, top_level: int) -> None:
        super().__init__()
        if not 0 < bottom_level < top_level:
            raise AssertionError((bottom_level, top_level))
        self.bottom_level, self.top_level = bottom_level, top_level
        self.in_levels = range(bottom_level, min(top_level, len(in_channels) - 1) + 1)
        width = top_level - bottom_level + 1
        self.out_channels = list(in_channels)
        self.out_channels[bottom_level: top_level + 1] = [out_channels] * width
        block = Conv2dNormActivation
        self.input_projections = _modules(len(self.in_levels),
                                          lambda i: block(in_channels[self.in_levels[i]], out_channels, 1))
        self.up_convs = _modules(len(self.in_levels) - 1, lambda _: block(out_channels, out_channels, 1))
        self.extra_downscalers = _modules(top_level - len(in_channels) + 1,
                                          lambda _: block(out_channels, out_channels, stride=2))
        self.out_convs = _modules(width, lambda _: block(out_channels, out_channels))

    def forward(self, inputs: Sequence[Tensor]) -> List[Tensor]:
        lo = self.in_levels.start
        proj = [p.forward_nhwc(ops.nhwc(inputs[lo + i])) for i, p in enumerate(self.input_projections)]
        pyramid = [proj[-1]]  # coarsest first while descending
        for i, conv in enumerate(self.up_convs):
            pyramid[i] = conv.forward_nhwc(pyramid[i])
            pyramid.append(ops.nearest_up2_add(pyramid[i], proj[-2 - i]))
        pyramid.reverse()
        for extra in self.extra_downscalers:
            pyramid.append(extra.forward_nhwc(pyramid[-1]))
        return _splice(inputs, self.bottom_level, self.top_level,
                       [conv.forward_nhwc(t) for conv, t in zip(self.out_convs, pyramid)])
