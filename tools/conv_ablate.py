"""Developer ablation (GPU box): where does the conv kernel spend its time on small / thin shapes?
dbg bits: 1 = no in-loop DMA, 2 = no ds_read/MFMA, 32 = no epilogue."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    lib.sihl_profile_enable(1)
    for _ in range(n):
        fn()
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
for name, N, H, W, Cin, Cout, K in [("L7 3x3", 32, 4, 4, 256, 256, 3), ("L6 3x3", 32, 8, 8, 256, 256, 3),
                                    ("L5 3x3", 32, 16, 16, 256, 256, 3), ("L4 3x3", 32, 32, 32, 256, 256, 3),
                                    ("L7 1x1", 32, 4, 4, 256, 256, 1), ("r3 1x1", 32, 32, 32, 1024, 256, 1),
                                    ("mlp", 1, 1, 174592, 256, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=dt)
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    for stats in (2,):
        line = f"{name} stats{stats}: "
        for mode, label in [(0, "default"), (1, "no-dma"), (2, "no-mfma"), (3, "neither"), (35, "neither-noepi"), (32, "noepi"), (64 + 3, "no-loop"), (64 + 35, "no-loop-noepi")]:
            lib.sihl_conv2d_debug(mode)
            t = timeit(lambda: ops.conv2d_raw(x, w, None, 1, K // 2, 1, act="relu", stats_mode=stats))
            line += f"{label} {t * 1e6:6.1f} | "
        lib.sihl_conv2d_debug(0)
        print(line, flush=True)
