"""Developer ablation (GPU box, tuning build): where does conv_pw_kernel spend a tile?
SIHL_HIP_LIB=tools/micro/libsihl_full_tuning.so python tools/pw_ablate.py     dbg bits: 2 = no ds_read/MFMA, 32 = no epilogue"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
NB = 6


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
    return ts[len(ts) // 2] * 1e3


_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(50):
    _w @ _w
for name, N, H, W, Cin, Cout in [("r1 64>256", 32, 128, 128, 64, 256), ("r2 128>512", 32, 64, 64, 128, 512),
                                 ("mlp 256>256", 1, 1, 174592, 256, 256), ("r3 256>1024", 32, 32, 32, 256, 1024)]:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(Cout, 1, 1, Cin, device=dev, dtype=dt) * 0.05
    for stats in (0, 1):
        line = f"{name:12s} stats{stats}: "
        for pw in (0, 1):
            lib.sihl_conv2d_pw_enable(pw)
            for mode, label in [(0, "all"), (2, "no-mfma"), (32, "no-epi"), (34, "loads only")]:
                lib.sihl_conv2d_debug(mode)
                t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 0, 1, act=None, stats_mode=stats))
                line += f"{'pw' if pw else 'tile'} {label} {t:6.1f} | "
            lib.sihl_conv2d_debug(0)
        lib.sihl_conv2d_pw_enable(1)
        print(line, flush=True)
