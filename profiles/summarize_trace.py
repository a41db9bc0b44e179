#!/usr/bin/env python3
"""Steady-state per-step kernel summary from a rocprofv3 --kernel-trace CSV of `bench.py`.

The first warm-up iteration contains MIOpen's JIT / naive fallback kernels, so totals are taken only over the
last N steps, delimited by a kernel that runs exactly once per training step (od_anchors_kernel)."""
import collections
import csv
import sys

path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "od_anchors_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-nsteps - 1], marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def cat(n):
    mine = ("conv_igemm", "conv_wgrad", "colsum", "norm_bwd", "affine_act", "layernorm", "fuse_", "blur_", "up2_",
            "bn_finalize", "bn_eval", "weight_flip", "wgrad_reduce", "topk", "gather_rows", "od_", "fusion_wgrad",
            "resize", "nearest", "uafm", "softmax_max", "ce_resize", "maxpool3x3s2", "conv_splitk", "conv_add", "weight_prepare",
            "iseg_mask", "conv_small", "conv_pyr", "conv_halo", "mlp_rows", "mlp_fused", "bn_stats", "od_decode", "colsum", "stem_")
    if any(k in n for k in mine):
        return "sihl_hip"
    if n.startswith("MIOpen") or "igemm_" in n or "ck::" in n or "_ZN2ck" in n or "SubTensor" in n or "naive_conv" in n \
            or "Op2dTensor" in n or "gridwise" in n.lower() or "batched_transpose" in n:
        return "MIOpen/CK (backbone convs + BN)"
    return "torch ATen / rocclr"


cats = collections.defaultdict(lambda: [0, 0.0])
for n, (c, ms) in agg.items():
    k = cats[cat(n)]
    k[0] += c
    k[1] += ms
wall = (t1 - t0) / 1e6 / nsteps
print(f"steps analysed: {nsteps}; wall per step {wall:.2f} ms; kernel time per step {sum(v[1] for v in agg.values()) / nsteps:.2f} ms")
for k, (c, ms) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f"  {ms / nsteps:8.2f} ms/step {c / nsteps:8.0f} launches/step  {k}")
print("top kernels:")
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"  {ms / nsteps:8.3f} ms/step {c / nsteps:7.1f}/step avg {ms / c * 1e3:8.1f} us  [{cat(n)[:8]}] {n[:110]}")
