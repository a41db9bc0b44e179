"""Developer benchmark (GPU box): persistent pointwise kernel (csrc/conv_pw.hip) against the one-tile-per-workgroup
kernel on the HBM-bound 1x1 shapes of the ResNet50 + BiFPN + detection-head step, rotating inputs (> 256 MB)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
NB = 6


def timeit(fn, n=30):
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
    return ts[len(ts) // 2] * 1e3  # us


_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(50):
    _w @ _w
SHAPES = [("r1 64>256", 32, 128, 128, 64, 256), ("r1 64>64... n/a", 0, 0, 0, 0, 0), ("r1 256>64", 32, 128, 128, 256, 64),
          ("r2 128>512", 32, 64, 64, 128, 512), ("r2 256>128", 32, 128, 128, 256, 128), ("r2 256>512", 32, 64, 64, 256, 512),
          ("mlp 256>256", 1, 1, 174592, 256, 256), ("r3 256>1024", 32, 32, 32, 256, 1024),
          ("P3 256>256", 32, 64, 64, 256, 256), ("P4 256>256", 32, 32, 32, 256, 256)]
for name, N, H, W, Cin, Cout in SHAPES:
    if not N:
        continue
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(Cout, 1, 1, Cin, device=dev, dtype=dt) * 0.05
    mb = (N * H * W * (Cin + Cout) + Cin * Cout) * 2 / 1e6
    gf = 2.0 * N * H * W * Cin * Cout / 1e9
    line = f"{name:12s} {mb:6.1f} MB {gf:6.1f} GF: "
    for stats in (0, 1):
        ts = []
        for mode in (0, 1, 9):  # tile kernel | persistent, exact ring waits | persistent, compiler stores + conservative waits
            lib.sihl_conv2d_pw_enable(mode)
            ts.append(timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 0, 1, act=None, stats_mode=stats)))
        lib.sihl_conv2d_pw_enable(1)
        line += f"stats{stats}: tile {ts[0]:6.1f} us ({mb / ts[0]:4.2f} TB/s) -> pw {ts[1]:6.1f} us ({mb / ts[1]:4.2f} TB/s, {gf / ts[1] * 1e3:5.0f} TF) conservative {ts[2]:6.1f} | "
    print(line, flush=True)
