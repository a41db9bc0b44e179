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
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.sihl_wgrad_stamps.argtypes = [ctypes.c_void_p]
for name, N, H, W in (("P3", 32, 64, 64), ("P4", 32, 32, 32)):
    x = torch.randn(N, H, W, 256, device=dev, dtype=dt)
    dy = torch.randn(N, H, W, 256, device=dev, dtype=dt)
    for _ in range(10):
        ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    assert lib.sihl_wgrad_stamps(buf.data_ptr()) == 0
    ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    torch.cuda.synchronize()
    lib.sihl_wgrad_stamps(None)
    t = buf[:80].cpu().reshape(16, 5).tolist()
    ns = t[0][4]
    tot = sum(r[0] for r in t) / 16
    print(f"{name}: {ns} stages of 64 pixels, K loop {tot:.0f} cycles per wave = {tot / max(1, ns):.0f} per stage (2 048 matrix cycles of the SIMD)")
    for wv in (0, 3, 7, 8, 12, 15):
        r = t[wv]
        print(f"   wave {wv:2d}: multiply + issue {100 * r[1] / r[0]:4.1f} % | own DMA wait {100 * r[2] / r[0]:4.1f} % | barrier {100 * r[3] / r[0]:4.1f} %")
