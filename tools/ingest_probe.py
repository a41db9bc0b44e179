"""Developer probe (GPU box, SIHL_HIP_LIB = a `make TUNING=1`-style build): how much of a pyramid conv's time is the
LDS-DMA ingest of its two operands?  Timing ablations of sihl_conv2d_debug (results invalid): 128 = the input tile is
fetched for the first tap of each channel chunk only (what a halo-resident input tile would fetch), 256 = the same for the
weights, 1 = no DMA in the loop at all, 2 = no ds_read / MFMA."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
SHAPES = [("L3 3x3", 32, 64, 64, 256, 256, 3), ("L4 3x3", 32, 32, 32, 256, 256, 3), ("L5 3x3", 32, 16, 16, 256, 256, 3),
          ("L6 3x3", 32, 8, 8, 256, 256, 3), ("L7 3x3", 32, 4, 4, 256, 256, 3)]
NB = 8


def timeit(fn, n=24):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    lib.sihl_profile_enable(1)
    for i in range(n):
        fn(i % NB)
    torch.cuda.synchronize()
    lib.sihl_profile_enable(0)
    cnt = lib.sihl_profile_records(0, _C.BF16, None, 0)
    buf = (ctypes.c_double * (3 * cnt))()
    lib.sihl_profile_records(0, _C.BF16, buf, cnt)
    ts = sorted(buf[3 * i] for i in range(cnt))
    return ts[len(ts) // 2] * 1e-3


_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(100):
    _w @ _w
torch.cuda.synchronize()
for name, N, H, W, Cin, Cout, K in SHAPES:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    line = f"{name:8s}"
    for label, mode in (("all", 0), ("A first tap only", 128), ("B first tap only", 256), ("A+B first tap", 384), ("no DMA", 1),
                        ("no MFMA", 2), ("no MFMA, A first tap", 130)):
        lib.sihl_conv2d_debug(mode)
        t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", post=(sc, sh)))
        line += f" | {label} {t * 1e6:6.1f}"
    lib.sihl_conv2d_debug(0)
    print(line, flush=True)
