import ctypes, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops
lib = _C.lib(); dev = "cuda"; dt = torch.bfloat16
NB = 6
def timeit(fn, n=24):
    for i in range(NB): fn(i)
    torch.cuda.synchronize(); lib.sihl_profile_enable(1)
    for i in range(n): fn(i % NB)
    torch.cuda.synchronize(); lib.sihl_profile_enable(0)
    cnt = lib.sihl_profile_records(0, _C.BF16, None, 0); buf = (ctypes.c_double * (3 * cnt))()
    lib.sihl_profile_records(0, _C.BF16, buf, cnt); ts = sorted(buf[3 * i] for i in range(cnt)); return ts[len(ts) // 2] * 1e3
_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(100): _w @ _w
for name, N, H, W, Cin, Cout, K in [("L3 3x3", 32, 64, 64, 256, 256, 3), ("lat3 512>256", 32, 64, 64, 512, 256, 1), ("mlp", 1, 1, 174592, 256, 256, 1),
                                    ("r1 64>256", 32, 128, 128, 64, 256, 1), ("r2 128>512", 32, 64, 64, 128, 512, 1)]:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    ref = None
    line = f"{name:14s}"
    for ov in (0, 1, 2):
        lib.sihl_conv2d_tile_override(1280 if ov else 0)
        lib.sihl_conv2d_nbuf_override(ov)
        t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, K // 2, 1, act=None, stats_mode=1))
        y, _ = ops.conv2d_raw(xs[0], w, None, 1, K // 2, 1, act=None, stats_mode=1)
        if ref is None: ref = y
        line += f" | ov{ov} {t:6.1f} us {'ok' if torch.equal(ref, y) else 'DIFF'}"
    lib.sihl_conv2d_nbuf_override(0)
    lib.sihl_conv2d_tile_override(0)
    print(line, flush=True)
