import ctypes, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops
lib = _C.lib(); dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize(); lib.sihl_profile_enable(1)
    for _ in range(n): fn()
    torch.cuda.synchronize(); lib.sihl_profile_enable(0)
    cnt = lib.sihl_profile_records(0, _C.BF16, None, 0); buf = (ctypes.c_double * (3 * cnt))()
    lib.sihl_profile_records(0, _C.BF16, buf, cnt); ts = sorted(buf[3 * i] for i in range(cnt)); return ts[len(ts) // 2] * 1e3
_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(100): _w @ _w
for name, N, H, W, Cin, Cout, K in [("L7 3x3", 32, 4, 4, 256, 256, 3), ("L6 3x3", 32, 8, 8, 256, 256, 3), ("L5 3x3", 32, 16, 16, 256, 256, 3), ("lat7 1x1 2048", 32, 4, 4, 2048, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device=dev, dtype=dt); w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    line = name
    for on in (0, 1):
        lib.sihl_conv2d_splitk_enable(on)
        t = timeit(lambda: ops.conv2d_raw(x, w, None, 1, K // 2, 1, act="relu", stats_mode=2))
        line += f" | splitk={on} {t:.1f} us"
    lib.sihl_conv2d_splitk_enable(1)
    print(line, flush=True)
