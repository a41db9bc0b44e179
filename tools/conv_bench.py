"""Developer micro-benchmark (GPU box): the matrix-core kernels on the flagship shapes."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev = "cuda"
dt = torch.bfloat16
SHAPES = [("L3 3x3", 32, 64, 64, 256, 256, 3), ("L4 3x3", 32, 32, 32, 256, 256, 3), ("L5 3x3", 32, 16, 16, 256, 256, 3),
          ("L6 3x3", 32, 8, 8, 256, 256, 3), ("L7 3x3", 32, 4, 4, 256, 256, 3),
          ("lat3 1x1", 32, 64, 64, 512, 256, 1), ("lat5 1x1", 32, 16, 16, 2048, 256, 1),
          ("mlp 1x1", 1, 1, 174592, 256, 256, 1), ("cls 1x1", 1, 1, 3200, 256, 80, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


modes = [("auto", 0), ("bn256", 128), ("bn128", 1280), ("bn64", 64)]
for name, N, H, W, Cin, Cout, K in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=dt)
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    dy = torch.randn(N, H, W, Cout, device=dev, dtype=dt)
    flops = 2.0 * N * H * W * Cin * Cout * K * K
    line = f"{name:10s} {flops/1e9:8.1f} GF "
    for mname, flag in modes:
        _C.lib().sihl_conv2d_tile_override(flag)
        t = timeit(lambda: ops.conv2d_raw(x, w, None, 1, K // 2, 1, act="relu", stats_mode=2))
        line += f"| fwd[{mname}] {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s "
    _C.lib().sihl_conv2d_tile_override(0)
    t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, K, K, 1, K // 2, 1))
    line += f"| wgrad {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s"
    print(line, flush=True)
