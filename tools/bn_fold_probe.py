"""Developer probe (GPU box): what a loader-side BatchNorm fold would have to beat, layer by layer (round-3 review item 5b).
For the 16 conv2 -> conv3 pairs of the ResNet50 trunk at bs 32, 512^2, bf16: the pointwise conv3 through the LDS-DMA loader
(shipped) and through the register-staged loader (the only one that could apply scale / shift / ReLU on load: measured WITHOUT
that arithmetic, so a lower bound of its cost), stats epilogue on as in training, against the `affine_act` pass over conv3's
input that the fold would remove.  The fold wins a layer only if  t(register-staged) - t(LDS-DMA) < t(affine_act)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {"dma": 0.0, "reg": 0.0, "aff": 0.0}
print("layer (blocks)             width -> out   map     conv3 LDS-DMA   register-staged   affine_act on its input   fold wins?")
for name, blocks, width, hw in (("layer1", 3, 64, 128), ("layer2", 4, 128, 64), ("layer3", 6, 256, 32), ("layer4", 3, 512, 16)):
    x = torch.randn(32, hw, hw, width, device=dev, dtype=dt)
    w = torch.randn(4 * width, 1, 1, width, device=dev, dtype=dt) * 0.05
    scale = torch.rand(width, device=dev) + 0.5
    shift = torch.randn(width, device=dev)
    lib.sihl_conv2d_force_register_staging(0)
    t_dma = timed(lambda: ops.conv2d_raw(x, w, stats_mode=1))
    lib.sihl_conv2d_force_register_staging(1)
    t_reg = timed(lambda: ops.conv2d_raw(x, w, stats_mode=1))
    lib.sihl_conv2d_force_register_staging(0)
    t_aff = timed(lambda: ops.affine_act(x, scale, shift, "relu"))
    tot["dma"] += blocks * t_dma
    tot["reg"] += blocks * t_reg
    tot["aff"] += blocks * t_aff
    print(f"{name} x{blocks}   {width:4d} -> {4 * width:4d}   {hw:3d}^2   {t_dma:8.1f} us   {t_reg:12.1f} us   {t_aff:14.1f} us            "
          f"{'yes' if t_reg - t_dma < t_aff else 'NO'} ({t_reg - t_dma:+.1f} vs -{t_aff:.1f})")
print(f"per step, forward, all 16 pairs: LDS-DMA {tot['dma'] / 1e3:.3f} ms, register-staged {tot['reg'] / 1e3:.3f} ms "
      f"(+{(tot['reg'] - tot['dma']) / 1e3:.3f}), affine_act passes removed {tot['aff'] / 1e3:.3f} ms")
