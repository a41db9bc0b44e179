"""Developer probe (GPU box): tile-shape overrides of the LDS-DMA conv on the pyramid shapes of the north-star forward
(inference epilogue: ReLU + folded BatchNorm), median launch time by HIP events (sihl profiler)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
SHAPES = [("L3 3x3", 32, 64, 64, 256, 256, 3), ("L4 3x3", 32, 32, 32, 256, 256, 3), ("L5 3x3", 32, 16, 16, 256, 256, 3),
          ("L6 3x3", 32, 8, 8, 256, 256, 3), ("L7 3x3", 32, 4, 4, 256, 256, 3), ("mlp 1x1", 1, 1, 174592, 256, 256, 1),
          ("lat3 1x1", 32, 64, 64, 512, 256, 1), ("r2 128>512", 32, 64, 64, 128, 512, 1), ("r1 64>256", 32, 128, 128, 64, 256, 1)]
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
    flops = 2.0 * N * H * W * Cin * Cout * K * K
    line = f"{name:10s} {flops / 1e9:7.1f} GF"
    for tname, tile, nb in (("auto", 0, 0), ("256x256w16", 256, 0), ("256x256w8", 2568, 0), ("256x256w16", 256, 0),
                            ("256x256w8", 2568, 0), ("128x256w8/2", 2562, 0), ("128x256w8/3", 2563, 0),
                            ("128x128/1", 1280, 1), ("128x128/2", 1280, 2), ("128x64/2", 64, 2), ("128x64/4", 64, 4)):
        lib.sihl_conv2d_tile_override(tile)
        lib.sihl_conv2d_nbuf_override(nb)
        try:
            t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act="relu", post=(sc, sh)))
            line += f" | {tname} {t * 1e6:6.1f}"
        except Exception:  # noqa: BLE001
            line += f" | {tname} ERR"
    lib.sihl_conv2d_tile_override(0)
    lib.sihl_conv2d_nbuf_override(0)
    print(line, flush=True)
