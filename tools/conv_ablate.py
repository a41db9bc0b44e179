"""Developer ablation (GPU box): where does a K stage of the conv kernel spend its time?"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, N, H, W in [("L3", 32, 64, 64), ("L5", 32, 16, 16)]:
    x = torch.randn(N, H, W, 256, device=dev, dtype=dt)
    w = torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.05
    for bm in (128, 256):
        _C.lib().sihl_conv2d_tile_override(bm)
        line = f"{name} bm{bm}: "
        for mode, label in [(0, "default"), (3, "neither"), (35, "neither-noepi"), (32, "noepi")]:
            _C.lib().sihl_conv2d_debug(mode)
            t = timeit(lambda: ops.conv2d_raw(x, w, None, 1, 1, 1))
            line += f"{label} {t*1e6:7.1f} us | "
        _C.lib().sihl_conv2d_debug(0)
        print(line, flush=True)
_C.lib().sihl_conv2d_tile_override(0)
