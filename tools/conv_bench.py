"""Developer micro-benchmark (GPU box): the LDS-DMA conv kernel on the shapes of the flagship step, per tile
width and LDS stage count."""
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
          ("mlp 1x1", 1, 1, 174592, 256, 256, 1), ("cls 1x1", 1, 1, 3200, 256, 80, 1),
          ("r1 1x1 64>256", 32, 128, 128, 64, 256, 1), ("r1 1x1 256>64", 32, 128, 128, 256, 64, 1),
          ("r1 3x3 64", 32, 128, 128, 64, 64, 3), ("r2 3x3 128", 32, 64, 64, 128, 128, 3),
          ("r2 1x1 128>512", 32, 64, 64, 128, 512, 1), ("r2 1x1 512>128", 32, 64, 64, 512, 128, 1),
          ("r3 1x1 1024>256", 32, 32, 32, 1024, 256, 1), ("r3 1x1 256>1024", 32, 32, 32, 256, 1024, 1),
          ("r3 3x3 256", 32, 32, 32, 256, 256, 3), ("r4 3x3 512", 32, 16, 16, 512, 512, 3),
          ("r4 1x1 2048>512", 32, 16, 16, 2048, 512, 1), ("r4 1x1 512>2048", 32, 16, 16, 512, 2048, 1)]
lib = _C.lib()
NBUF = 8  # rotate operands: keep the Infinity Cache from serving repeats of small tensors


import ctypes  # noqa: E402


def timeit(fn, n=24):
    """Median GPU time of the conv launch itself (HIP events around the launch, sihl profiler), seconds."""
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


# clock warm-up
_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
for name, N, H, W, Cin, Cout, K in SHAPES:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NBUF)]
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    flops = 2.0 * N * H * W * Cin * Cout * K * K
    line = f"{name:16s} {flops / 1e9:7.1f} GF "
    for nb in (2, 3, 4, 0):
        lib.sihl_conv2d_nbuf_override(nb)
        t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", stats_mode=2))
        line += f"| auto/nbuf{nb} {t * 1e6:7.1f} us {flops / t / 1e12:6.0f} TF "
    for tname, flag in (("bn128", 1280), ("bn64", 64)):
        if Cout <= 128:
            continue
        lib.sihl_conv2d_tile_override(flag)
        for nb in (2, 4):
            lib.sihl_conv2d_nbuf_override(nb)
            t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", stats_mode=2))
            line += f"| {tname}/nbuf{nb} {t * 1e6:7.1f} us "
    lib.sihl_conv2d_tile_override(0)
    lib.sihl_conv2d_nbuf_override(0)
    print(line, flush=True)
