"""Developer probe (GPU box; SIHL_HIP_LIB = a library whose conv_halo.hip was compiled with -DSIHL_HALO_STAMPS): where the waves
of workgroup 0 of the halo-resident P3 conv spend their K loop - s_memtime sums per wave: waiting for its own DMA pieces,
waiting at the stage barrier, multiplying (incl. the next stage's DMA issue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C  # noqa: E402
from sihl_amd.ops import _p, _stream  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
N, H, W, C = 32, 64, 64, 256
x = torch.randn(N, H, W, C, device=dev, dtype=dt)
w = torch.randn(C, 3, 3, C, device=dev, dtype=dt) * 0.02
out = torch.empty_like(x)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
ws = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
for mode, name in ((3, "barrier in front of the stage"), (2, "barrier before the last tap's multiplies (default)")):
    lib.sihl_conv2d_halo_enable(mode)
    for _ in range(30):
        rc = lib.sihl_conv2d_fwd_ws(_p(x), _p(w), None, _p(out), N, H, W, C, C, 3, 3, 1, 1, 1, _C.BF16, _C.ACT["relu"], None, None,
                                    _p(sc), _p(sh), 0, None, 0, 0, _p(ws), ws.numel() * 8, _stream())
        assert rc == 0, rc
    torch.cuda.synchronize()
    t = ws[:64].cpu().reshape(16, 4).tolist()
    tot = sum(r[0] for r in t) / 16
    print(f"{name}: K loop {tot:.0f} cycles per wave (24 stages: {tot / 24:.0f} per stage, 3 072 of them matrix cycles of the SIMD)")
    print("   wave: total | own DMA wait | barrier wait | multiply + issue")
    pe = ws[64:96].cpu().reshape(16, 2).tolist()
    print(f"   prologue (entry -> K loop) {sum(r[0] for r in pe) / 16:.0f} cycles, epilogue (K loop end -> stores issued) {sum(r[1] for r in pe) / 16:.0f} cycles")
    for wv in (0, 3, 7, 8, 12, 15):
        r = t[wv]
        print(f"   {wv:4d}: {r[0]:6d} | {r[1]:6d} ({100 * r[1] / r[0]:4.1f} %) | {r[2]:6d} ({100 * r[2] / r[0]:4.1f} %) | {r[3]:6d} ({100 * r[3] / r[0]:4.1f} %)")
lib.sihl_conv2d_halo_enable(1)
