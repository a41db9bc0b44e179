"""ConvNormAct / SequentialConvBlocks / Conv2dNormActivation on the HIP path.

Mirrors src/sihl/layers/convblocks.py:37-117 of the reference (and torchvision's
ops.Conv2dNormActivation used by fpn.py:26-37 / heads/object_detection.py:52-55): same constructor
arguments, same child-module indices and therefore the same ``state_dict`` keys and shapes, so a
reference checkpoint loads key-for-key.  The children (nn.Conv2d / nn.BatchNorm2d) are parameter
containers only - ``forward`` runs the fused HIP kernels in ``sihl_amd.ops``.
"""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from sihl_amd import ops

_ACT_MODULES = {
    "relu": lambda: nn.ReLU(inplace=True),
    "silu": lambda: nn.SiLU(inplace=True),
    "sigmoid": nn.Sigmoid,
    "softplus": nn.Softplus,
    "softmax": lambda: nn.Softmax(dim=1),
}


def _vec(dtype) -> int:
    return 8 if dtype == torch.bfloat16 else 4


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _ConvBlock(nn.Sequential):
    """Shared forward for conv->act->norm ("act_norm") and conv->norm->act ("norm_act")."""

    order = "act_norm"
    act: Optional[str] = None

    def _parts(self):
        conv = self[0]
        norm = next((m for m in self if isinstance(m, nn.BatchNorm2d)), None)
        return conv, norm

    def forward_nhwc(self, x: Tensor, defer=None) -> Tensor:
        """defer: an ops.DeferredAffine when the only consumer of the result is a blur-pool that takes the BatchNorm
        affine over (AntialiasedDownscaler, BiFPNLayer)."""
        plan = self.__dict__.get("_sihl_plan")
        if plan is None:  # the block's structure is fixed after construction: checked once (~5 us x 140 blocks per forward)
            conv, bn = self._parts()
            if conv.groups != 1 or conv.padding_mode != "zeros":
                raise NotImplementedError("sihl_amd conv kernels cover groups=1, zero padding")
            (sh, sw), (ph, pw), (dh, dw) = conv.stride, conv.padding, conv.dilation
            if sh != sw or ph != pw or dh != dw:
                raise NotImplementedError("square stride / padding / dilation only")
            if bn is not None and (bn.momentum is None or not bn.track_running_stats or not bn.affine):
                raise NotImplementedError("BatchNorm2d with default affine / running-stat settings only")
            plan = self.__dict__["_sihl_plan"] = (conv, bn, sh, ph, dh)
        conv, bn, sh, ph, dh = plan
        weight, bias = conv.weight, conv.bias
        vec = _vec(x.dtype)
        cout = weight.shape[0]
        # zero-pad odd channel counts up to the 16-byte vector width (tiny convs only: UAFM 4->1, classifiers)
        if x.shape[-1] % vec:
            padc = _pad_to(x.shape[-1], vec) - x.shape[-1]
            x = F.pad(x, (0, padc))
            weight = F.pad(weight, (0, 0, 0, 0, 0, padc))
        if cout % vec:
            if bn is not None:
                raise NotImplementedError("normalised conv needs Cout to be a multiple of the vector width")
            weight = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, _pad_to(cout, vec) - cout))
            bias = F.pad(bias, (0, _pad_to(cout, vec) - cout)) if bias is not None else None
        if bn is not None:
            if self.training:
                ops.bump_counter(bn.num_batches_tracked)
            y = ops.conv_block(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, stride=sh,
                               pad=ph, dil=dh, act=self.act, order=self.order, training=self.training, eps=bn.eps,
                               momentum=bn.momentum, defer=defer if cout % vec == 0 else None)
        else:
            fused_act = self.act if self.act in (None, "relu", "silu", "sigmoid") else None
            y = ops.conv_block(x, weight, bias, None, None, None, None, stride=sh, pad=ph, dil=dh, act=fused_act)
        y = y if y.shape[-1] == cout else y[..., :cout]
        tail = [m for m in list(self)[1:] if isinstance(m, (nn.GroupNorm, nn.Softplus, nn.Softmax))]
        if tail:
            # the reference's rarer block variants (convblocks.py:76-85: softplus / softmax(dim=1) activations, GroupNorm):
            # the conv (+ bias) runs on the HIP kernel, these modules as device ops on the NCHW view of its output
            v = ops.nchw_view(y)
            for m in tail:
                v = m(v)
            y = ops.nhwc(v)
        return y

    def forward_fused_node(self, fuse, defer=None) -> Optional[Tensor]:
        """This block applied to a BiFPN fusion node, node and conv in ONE launch (ops.fused_node_conv_block; fuse = ("up2", a,
        b, wraw) or ("blur", a, b, c, wraw, a_affine)).  None when the block or the shapes are outside that kernel: the caller
        then computes the node itself and calls forward_nhwc."""
        if self.order != "act_norm" or self.act not in (None, "relu"):
            return None
        conv, bn = self._parts()
        if (bn is None or conv.groups != 1 or conv.padding_mode != "zeros" or conv.kernel_size != (3, 3) or conv.stride != (1, 1)
                or conv.padding != (1, 1) or conv.dilation != (1, 1) or bn.momentum is None or not bn.track_running_stats
                or not bn.affine or any(isinstance(m, (nn.GroupNorm, nn.Softplus, nn.Softmax)) for m in self)):
            return None
        y = ops.fused_node_conv_block(fuse, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      act=self.act, training=self.training, eps=bn.eps, momentum=bn.momentum, defer=defer)
        if y is not None and self.training:
            ops.bump_counter(bn.num_batches_tracked)
        return y

    def forward_nhwc_emit(self, x: Tensor, emit, write_y: bool = True):
        """Inference only: this block on ``x`` AND the fusion node that consumes its output, in one launch (conv_pyr.hip's
        epilogue computes the node for each workgroup's channel slice of the whole map).  emit = ("up2", b, wraw) ->
        softmax(wraw)_0 * bilinear_x2(y) + .._1 * b, or ("blur", b, c, wraw) -> softmax(wraw)_0 * blur_s2(y) + .._1 * b +
        .._2 * c.  Returns (y, node) - y None when write_y is False - or None when the block, the mode or the shapes are
        outside that kernel (the caller then runs the block and the node's own kernel)."""
        if self.training or torch.is_grad_enabled() or self.order != "act_norm" or self.act not in (None, "relu"):
            return None
        conv, bn = self._parts()
        if (bn is None or conv.groups != 1 or conv.padding_mode != "zeros" or conv.kernel_size != (3, 3) or conv.stride != (1, 1)
                or conv.padding != (1, 1) or conv.dilation != (1, 1) or not bn.track_running_stats or not bn.affine
                or any(isinstance(m, (nn.GroupNorm, nn.Softplus, nn.Softmax)) for m in self)):
            return None
        return ops.conv_block_emit(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, act=self.act,
                                   eps=bn.eps, emit=emit, write_y=write_y)

    def forward_nhwc_into(self, x: Tensor, out: Tensor, out_image_stride: int) -> bool:
        """Inference only: write this block's output for image n at ``out`` + n * out_image_stride elements (a slice of a
        larger buffer).  Returns False - and writes nothing - when the block is not the plain case the one-launch path
        covers (the caller then uses forward_nhwc)."""
        conv, bn = self._parts()
        vec = _vec(x.dtype)
        (sh, sw), (ph, pw), (dh, dw) = conv.stride, conv.padding, conv.dilation
        if (self.training or torch.is_grad_enabled() or conv.groups != 1 or conv.padding_mode != "zeros" or sh != sw
                or ph != pw or dh != dw or x.shape[-1] % vec or conv.weight.shape[0] % vec or not x.is_cuda
                or any(isinstance(m, (nn.GroupNorm, nn.Softplus, nn.Softmax)) for m in self)
                or (bn is not None and (not bn.track_running_stats or not bn.affine))):
            return False
        if bn is not None:
            ops.conv_block_into(out, out_image_stride, x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                bn.running_var, stride=sh, pad=ph, dil=dh, act=self.act, order=self.order, eps=bn.eps)
        else:
            ops.conv_block_into(out, out_image_stride, x, conv.weight, conv.bias, None, None, None, None, stride=sh, pad=ph,
                                dil=dh, act=self.act)
        return True

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class ConvNormAct(_ConvBlock):
    """conv -> activation -> norm (note the order), reference convblocks.py:37-87."""

    order = "act_norm"

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, groups=1, padding=None,
                 norm="batch", act="relu", bias=None):
        pad = padding or ((kernel_size - 1) // 2 * dilation)  # padding=0 falls through, as in the reference
        use_bias = (norm is None) if bias is None else bool(bias)
        mods: List[nn.Module] = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, pad, dilation=dilation,
                                           groups=groups, bias=use_bias)]
        if act is not None:
            if act not in _ACT_MODULES:
                raise ValueError(f"unknown activation {act!r}")
            mods.append(_ACT_MODULES[act]())
        if norm == "batch":
            mods.append(nn.BatchNorm2d(out_channels))
        elif norm == "group":
            mods.append(nn.GroupNorm(in_channels // 8, out_channels))  # (the reference's group count, convblocks.py:85)
        elif norm is not None:
            raise ValueError(f"unknown norm {norm!r}")
        super().__init__(*mods)
        self.act = act


class Conv2dNormActivation(_ConvBlock):
    """torchvision.ops.Conv2dNormActivation: conv(bias=False) -> BatchNorm -> activation."""

    order = "norm_act"

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1,
                 norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, dilation=1, inplace=True, bias=None):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        mods: List[nn.Module] = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                           dilation=dilation, groups=groups, bias=bias)]
        if norm_layer is not None:
            if norm_layer is not nn.BatchNorm2d:
                raise NotImplementedError("BatchNorm2d only")
            mods.append(norm_layer(out_channels))
        act = None
        if activation_layer is not None:
            act = {nn.ReLU: "relu", nn.SiLU: "silu", nn.Sigmoid: "sigmoid"}.get(activation_layer)
            if act is None:
                raise NotImplementedError(f"{activation_layer} is outside the HIP hot path")
            mods.append(activation_layer(inplace=inplace) if activation_layer is not nn.Sigmoid else nn.Sigmoid())
        super().__init__(*mods)
        self.act = act
        self.out_channels = out_channels


class SequentialConvBlocks(nn.Sequential):
    """reference convblocks.py:96-117"""

    def __init__(self, in_channels, out_channels, num_layers, kernel_size=3, **kw):
        if num_layers <= 0:
            super().__init__(nn.Identity())
            return
        chans = [in_channels] + [out_channels] * num_layers
        super().__init__(*[ConvNormAct(chans[i], chans[i + 1], kernel_size=kernel_size, **kw)
                           for i in range(num_layers)])

    def forward_nhwc(self, x: Tensor) -> Tensor:
        for m in self:
            if not isinstance(m, nn.Identity):
                x = m.forward_nhwc(x)
        return x

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))
