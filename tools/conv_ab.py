"""Developer probe (GPU box): default-dispatch conv times on the flagship shapes, one line (for same-box A/B of two
builds: SIHL_HIP_LIB=<alt .so> python tools/conv_ab.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

lib = _C.lib()
dev, dt, NBUF = "cuda", torch.bfloat16, 8
SHAPES = [("L3", 32, 64, 64, 256, 256, 3), ("L4", 32, 32, 32, 256, 256, 3), ("L5", 32, 16, 16, 256, 256, 3),
          ("L6", 32, 8, 8, 256, 256, 3), ("lat3", 32, 64, 64, 512, 256, 1), ("lat5", 32, 16, 16, 2048, 256, 1),
          ("mlp", 1, 1, 174592, 256, 256, 1), ("r1e", 32, 128, 128, 64, 256, 1), ("r1r", 32, 128, 128, 256, 64, 1),
          ("r1c", 32, 128, 128, 64, 64, 3), ("r2c", 32, 64, 64, 128, 128, 3), ("r2e", 32, 64, 64, 128, 512, 1),
          ("r3r", 32, 32, 32, 1024, 256, 1), ("r3e", 32, 32, 32, 256, 1024, 1), ("r3c", 32, 32, 32, 256, 256, 3),
          ("r4c", 32, 16, 16, 512, 512, 3), ("r4r", 32, 16, 16, 2048, 512, 1), ("r4e", 32, 16, 16, 512, 2048, 1)]


def timeit(fn, n=24):
    for i in range(NBUF):
        fn(i)
    torch.cuda.synchronize()
    lib.sihl_profile_enable(1)
    for i in range(n):
        fn(i % NBUF)
    torch.cuda.synchronize()
    lib.sihl_profile_enable(0)
    cnt = lib.sihl_profile_records(0, _C.BF16, None, 0)
    buf = (ctypes.c_double * (3 * cnt))()
    lib.sihl_profile_records(0, _C.BF16, buf, cnt)
    ts = sorted(buf[3 * i] for i in range(cnt))
    return ts[len(ts) // 2] * 1e-3


_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
out = []
for name, N, H, W, Cin, Cout, K in SHAPES:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NBUF)]
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", stats_mode=2))
    out.append(f"{name} {t * 1e6:.1f}")
print(" | ".join(out), flush=True)
