"""Developer probe (GPU box, SIHL_HIP_LIB = a -DSIHL_MLP_STAMPS build): in-kernel timeline of workgroup 0 of the whole-MLP
kernel (s_memtime marks: 100 MHz constant clock -> 10 ns per tick)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402
from sihl_amd.heads import mlp as mlp_mod  # noqa: E402

dev = "cuda"
lib = _C.lib()
lib.sihl_mlp_stages(2)
torch.manual_seed(0)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
lib.sihl_mlp_stamps(buf.data_ptr())
for rows, cout in ((3200, 80), (174592, 1)):
    m = mlp_mod.MLP(256, [256] * 4 + [cout], norm_layer=torch.nn.LayerNorm, activation_layer=torch.nn.SiLU).to(dev).eval()
    x = torch.randn(rows, 256, device=dev).bfloat16()
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
    names = {0: "start", 1: "X + first weight stages issued", 2: "landed, barrier"}
    for l in range(5):
        names[3 + 4 * l] = f"layer {l}: K loop done"
        names[4 + 4 * l] = f"layer {l}: z written, barrier"
        names[5 + 4 * l] = f"layer {l}: rows normalised"
        names[6 + 4 * l] = f"layer {l}: barrier"
    names[4 + 16] = "output stored"
    print(f"rows {rows} -> {cout}: workgroup 0, ticks of s_memtime (raw deltas; 100 MHz if constant clock)")
    last = None
    for i in range(0, 21):
        if i not in names or (i > 19 and i != 20):
            continue
        if i in (21, 22):
            continue
        if t[i] == 0:
            continue
        d = 0 if last is None else t[i] - last
        print(f"  {i:2d} {names[i]:34s} +{d:7d}   total {t[i] - t[0]:8d}")
        last = t[i]
