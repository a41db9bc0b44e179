"""BlurPool2d / AntialiasedDownscaler / Interpolate / SimpleUpscaler on the HIP path
(reference src/sihl/layers/pooling.py:7-30, scalers.py:16-56)."""
import numpy as np
import torch
from torch import Tensor, nn

from sihl_amd import ops
from sihl_amd.layers.convblocks import ConvNormAct


class BlurPool2d(nn.Module):
    """Reflect-pad + depthwise binomial filter; the HIP kernel covers the hot-path case
    (kernel_size=3, stride=2).  The ``kernel`` buffer is kept for state_dict parity only."""

    def __init__(self, in_channels: int, kernel_size: int = 3, stride: int = 1):
        super().__init__()
        self.in_channels, self.kernel_size, self.stride = in_channels, kernel_size, stride
        taps = np.ones(1)
        for _ in range(kernel_size - 1):
            taps = np.convolve(taps, [0.5, 0.5])
        t = torch.tensor(taps.astype(np.float32))
        self.register_buffer("kernel", torch.outer(t, t)[None, None].repeat(in_channels, 1, 1, 1))

    def forward_nhwc(self, x: Tensor, a_affine=None) -> Tensor:
        if self.kernel_size != 3 or self.stride != 2:
            raise NotImplementedError("HIP blur-pool covers kernel_size=3, stride=2 (the neck's downscaler)")
        return ops.blur_fuse(x, a_affine=a_affine)

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class AntialiasedDownscaler(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, **kw):
        super().__init__(ConvNormAct(in_channels, out_channels, kernel_size, **kw),
                         BlurPool2d(out_channels, stride=2))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        affine = ops.DeferredAffine()  # training: the conv block's BatchNorm affine rides in the blur (one pass less)
        return self[1].forward_nhwc(self[0].forward_nhwc(x, defer=affine), a_affine=affine)

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class Interpolate(nn.Module):
    """Bilinear resize, align_corners=False.  scale=2 runs the fused x2 kernel; an explicit ``size``
    runs the general bilinear kernel (SPPM)."""

    def __init__(self, scale=None, size=None, mode="bilinear"):
        super().__init__()
        self.scale, self.size, self.mode = scale, size, mode

    def forward_nhwc(self, x: Tensor) -> Tensor:
        if self.mode != "bilinear":
            raise NotImplementedError("bilinear only")
        if self.scale == 2 and self.size is None:
            return ops.up2(x)
        if self.size is not None:
            size = (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
            return ops.resize_bilinear(x, size)
        raise NotImplementedError("scale factors other than 2 are outside the HIP hot path")

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))


class SimpleUpscaler(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__(Interpolate(scale=2), ConvNormAct(in_channels, out_channels, kernel_size))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return self[1].forward_nhwc(self[0].forward_nhwc(x))

    def forward(self, x: Tensor) -> Tensor:
        return ops.nchw_view(self.forward_nhwc(ops.nhwc(x)))
