"""Developer probe (GPU box; SIHL_HIP_LIB = a library whose conv_wgrad.hip was compiled with -DSIHL_WGRAD_STAMPS): the K loop of
workgroup 0 of the LDS-DMA weight-gradient kernel on the P3 shape - s_memtime sums per wave: multiplying (+ DMA issue), waiting
for its own DMA pieces, waiting at the stage barrier."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

lib = ctypes.CDLL(os.environ["SIHL_HIP_LIB"])
_C.lib()
dev, dt = "cuda", torch.bfloat16
buf = torch.zeros(160, dtype=torch.int64, device=dev)
lib.sihl_wgrad_stamps.argtypes = [ctypes.c_void_p]
for name, N, H, W in (("P3", 32, 64, 64), ("P4", 32, 32, 32)):
    x = torch.randn(N, H, W, 256, device=dev, dtype=dt)
    dy = torch.randn(N, H, W, 256, device=dev, dtype=dt)
    for _ in range(10):
        ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per weight gradient (kernel + slab reduction), SIHL_WGRAD_WAVES={os.environ.get('SIHL_WGRAD_WAVES', '16')}")
    assert lib.sihl_wgrad_stamps(buf.data_ptr()) == 0
    ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    lib.sihl_wgrad_stamps(None)
    nwv = 8 if os.environ.get("SIHL_WGRAD_WAVES") == "8" else 16
    t = buf[:80].cpu().reshape(16, 5).tolist()[:nwv]
    ns = t[0][4]
    tot = sum(r[0] for r in t) / nwv
    print(f"{name}: {ns} stages of 64 pixels, K loop {tot:.0f} cycles per wave = {tot / max(1, ns):.0f} per stage (2 048 matrix cycles of the SIMD)")
    ex = buf[80:144].cpu().reshape(16, 4).tolist()[:nwv]
    clk = tot / max(1, sum(r[2] for r in ex) / nwv) * 100.0  # MHz: s_memtime ticks per 100 MHz s_memrealtime tick
    print(f"   in-kernel clock over the K loop {clk:.0f} MHz -> K loop {tot / clk:.1f} us; before the loop {sum(r[0] for r in ex) / nwv / clk:.1f} us, "
          f"slab stores (issue to vmcnt(0)) {sum(r[1] for r in ex) / nwv / clk:.1f} us")
    for wv in ((0, 3, 7, 8, 12, 15) if nwv == 16 else (0, 2, 4, 5, 6, 7)):
        r = t[wv]
        print(f"   wave {wv:2d}: multiply + issue {100 * r[1] / r[0]:4.1f} % | own DMA wait {100 * r[2] / r[0]:4.1f} % | barrier {100 * r[3] / r[0]:4.1f} %")
