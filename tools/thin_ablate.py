"""Developer ablation (GPU box): what bounds the HBM-bound pointwise convs (ResNet expansions / reductions, the MLP
linears)?  Needs the tuning build:  make -C tools/micro libsihl_full_tuning.so  and
SIHL_HIP_LIB=tools/micro/libsihl_full_tuning.so python tools/thin_ablate.py
dbg bits: 1 = no in-loop DMA, 2 = no ds_read/MFMA, 32 = no epilogue, 64 = no K loop at all."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
NB = 6  # rotating inputs: > 256 MB in flight, the Infinity Cache cannot serve repeats


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
    return ts[len(ts) // 2] * 1e3  # us


def ev_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


# yardsticks: write-only, read-only-ish (sum), copy on 512 MiB
big = torch.empty(256 << 20, device=dev, dtype=dt)
big2 = torch.empty_like(big)
t = ev_time(lambda: big.fill_(1.0))
print(f"fill 512 MiB: {t:7.1f} us  {big.numel() * 2 / t / 1e6:5.2f} TB/s written", flush=True)
t = ev_time(lambda: big2.copy_(big))
print(f"copy 512 MiB: {t:7.1f} us  {2 * big.numel() * 2 / t / 1e6:5.2f} TB/s read+written", flush=True)
del big, big2
_w = torch.randn(8192, 8192, device=dev, dtype=dt)
for _ in range(50):
    _w @ _w

SHAPES = [("r1 64>256", 32, 128, 128, 64, 256), ("r1 256>64", 32, 128, 128, 256, 64),
          ("r2 128>512", 32, 64, 64, 128, 512), ("r2 512>128", 32, 64, 64, 512, 128),
          ("mlp 256>256", 1, 1, 174592, 256, 256), ("r3 256>1024", 32, 32, 32, 256, 1024),
          ("r3 1024>256", 32, 32, 32, 1024, 256), ("r2 256>512", 32, 64, 64, 256, 512)]
for name, N, H, W, Cin, Cout in SHAPES:
    xs = [torch.randn(N, H, W, Cin, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(Cout, 1, 1, Cin, device=dev, dtype=dt) * 0.05
    mb = (N * H * W * (Cin + Cout) + Cin * Cout) * 2 / 1e6
    for stats in (0, 1):
        line = f"{name:12s} stats{stats} {mb:6.1f} MB: "
        for mode, label in [(0, "default"), (1, "no-dma"), (2, "no-mfma"), (32, "no-epi"), (34, "no-mfma-no-epi"), (64 + 3, "no-loop")]:
            lib.sihl_conv2d_debug(mode)
            t = timeit(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 0, 1, act=None, stats_mode=stats))
            line += f"{label} {t:6.1f}" + (f" ({mb / t:4.2f} TB/s)" if mode == 0 else "") + " | "
        lib.sihl_conv2d_debug(0)
        print(line, flush=True)
