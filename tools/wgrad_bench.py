"""Developer micro-benchmark (GPU box): weight-gradient kernels on the shapes of the flagship step (GPU-side times
of the main kernel from the sihl profiler; the split reduction is not included)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
SHAPES = [("r1 3x3 64", 32, 128, 128, 64, 64, 3, 1), ("r2 3x3 128", 32, 64, 64, 128, 128, 3, 1),
          ("r2 3x3 128 s2", 32, 128, 128, 128, 128, 3, 2), ("r1 1x1 64>256", 32, 128, 128, 64, 256, 1, 1),
          ("r1 1x1 256>64", 32, 128, 128, 256, 64, 1, 1), ("L3 3x3", 32, 64, 64, 256, 256, 3, 1),
          ("L4 3x3", 32, 32, 32, 256, 256, 3, 1), ("L5 3x3", 32, 16, 16, 256, 256, 3, 1),
          ("L6 3x3", 32, 8, 8, 256, 256, 3, 1), ("r3 3x3 256", 32, 32, 32, 256, 256, 3, 1),
          ("r4 3x3 512", 32, 16, 16, 512, 512, 3, 1), ("mlp", 1, 1, 174592, 256, 256, 1, 1)]


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.sihl_profile_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    lib.sihl_profile_enable(0)
    cnt = lib.sihl_profile_records(1, _C.BF16, None, 0)
    buf = (ctypes.c_double * (3 * cnt))()
    lib.sihl_profile_records(1, _C.BF16, buf, cnt)
    ts = sorted(buf[3 * i] for i in range(cnt))
    return ts[len(ts) // 2] * 1e-3


_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(100):
    _w @ _w
for name, N, H, W, Cin, Cout, K, st in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=dt)
    Ho, Wo = (H + 2 * (K // 2) - K) // st + 1, (W + 2 * (K // 2) - K) // st + 1
    dy = torch.randn(N, Ho, Wo, Cout, device=dev, dtype=dt)
    flops = 2.0 * N * Ho * Wo * Cin * Cout * K * K
    line = f"{name:16s} {flops / 1e9:7.1f} GF "
    for forced in (0, 1):
        lib.sihl_conv2d_wgrad_force_register_staging(forced)
        t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, K, K, st, K // 2, 1))
        line += f"| {'old' if forced else 'default'} {t * 1e6:7.1f} us {flops / t / 1e12:6.0f} TF "
    lib.sihl_conv2d_wgrad_force_register_staging(0)
    print(line, flush=True)
