#!/usr/bin/env python3
"""Per-step kernel summary of a rocprofv3 --kernel-trace CSV of the default `bench.py` run, taken over the HIP-graph
replays of the timed region (steps are delimited by od_anchors_kernel, which runs once per step; the trailing
eager profile steps and the warm-up are excluded)."""
import collections
import csv
import sys

path = sys.argv[1]
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # eager steps after the timed region
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # replays to average over
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "od_anchors_kernel" in r["Kernel_Name"]]
b = len(marks) - 1 - n_tail
a = b - n_steps
sel = rows[marks[a]:marks[b]]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = agg[r["Kernel_Name"]]
    k[0] += 1
    k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
wall = (int(rows[marks[b]]["Start_Timestamp"]) - int(rows[marks[a]]["Start_Timestamp"])) / 1e6 / n_steps
tot = sum(v[1] for v in agg.values()) / n_steps


def cat(n):
    mine = ("conv_igemm", "conv_wgrad", "colsum", "norm_bwd", "affine_act", "layernorm", "fuse_", "blur_", "up2_",
            "bn_finalize", "bn_eval", "weight_flip", "weight_prepare", "wgrad_reduce", "topk_rows", "gather_rows", "od_",
            "fusion_wgrad", "resize", "nearest", "uafm", "softmax_max", "ce_resize", "add_act")
    if any(k in n for k in mine):
        return "sihl_hip"
    if n.startswith("MIOpen") or "igemm_" in n or "ck::" in n or "SubTensor" in n or "naive_conv" in n or "Op2dTensor" in n:
        return "MIOpen (stem conv + BN)"
    return "torch ATen / rocclr"


cats = collections.defaultdict(lambda: [0, 0.0])
for n, (c, ms) in agg.items():
    k = cats[cat(n)]
    k[0] += c
    k[1] += ms
print(f"graph replays analysed: {n_steps}; wall/step {wall:.2f} ms; kernel time/step {tot:.2f} ms; "
      f"launches/step {sum(v[0] for v in agg.values()) / n_steps:.0f}")
for k, (c, ms) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f"  {ms / n_steps:8.2f} ms/step {c / n_steps:8.0f} launches/step  {k}")
print("top kernels:")
for name, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {ms / n_steps:8.3f} ms/step {c / n_steps:7.1f}/step avg {ms / c * 1e3:8.1f} us  [{cat(name)[:8]}] {name[:110]}")
