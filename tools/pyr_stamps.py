"""Developer probe (GPU box; SIHL_HIP_LIB = a library whose conv_pyr.hip was compiled with -DSIHL_PYR_STAMPS): in-kernel
timeline of workgroup 0 of the pyramid-top conv kernel - s_memtime (shader cycles) and s_memrealtime (100 MHz) marks."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = ctypes.CDLL(os.environ["SIHL_HIP_LIB"])
_C.lib()
buf = torch.zeros(16, dtype=torch.int64, device=dev)
lib.sihl_pyr_stamps.argtypes = [ctypes.c_void_p]
NAMES = ["start", "prologue issued (+ fused patch 0)", "stage 0 landed, barrier", "K loop done", "halves staged, barrier",
         "outputs stored", "end (statistics)"]
for name, N, W in (("L5 16x16", 32, 16), ("L6 8x8", 32, 8), ("L7 4x4", 32, 4)):
    x = torch.randn(N, W, W, 256, device=dev, dtype=dt)
    lo = torch.randn(N, W // 2, W // 2, 256, device=dev, dtype=dt)
    w = torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.02
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    w2 = torch.randn(2, device=dev)
    for label, call in (("eval plain", lambda: ops.pyr_conv_raw(w, x=x, act="relu", post=(sc, sh))),
                        ("train plain", lambda: ops.pyr_conv_raw(w, x=x, act="relu", stats_mode=2)),
                        ("eval up2-fused", (lambda: ops.pyr_conv_raw(w, fuse=("up2", lo, x, w2), act="relu", post=(sc, sh))) if W < 16 else None)):
        if call is None:
            continue
        lib.sihl_pyr_stamps(None)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        lib.sihl_pyr_stamps(buf.data_ptr())
        call()
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
        cyc, real = t[0::2], t[1::2]
        print(f"{name} {label}: total {(real[6] - real[0]) * 10} ns, {cyc[6] - cyc[0]} cycles "
              f"({(cyc[6] - cyc[0]) / max(1, (real[6] - real[0]) * 10) :.2f} GHz)")
        for i in range(1, 7):
            print(f"    {NAMES[i]:36s} +{(real[i] - real[i - 1]) * 10:6d} ns  +{cyc[i] - cyc[i - 1]:7d} cyc")
    sys.stdout.flush()
