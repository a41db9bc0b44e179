#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-graph`.

usage: pmc_summarize.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
Counters are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores.
Only dispatches of the steady-state steps are used (after the (warm-up)th od_anchors_kernel)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter, skip_steps):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    rows = [r for r in rows if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "od_anchors_kernel" in r["Kernel_Name"]]
    sel = rows[marks[skip_steps]:marks[-1]] if len(marks) > skip_steps + 1 else rows
    steps = max(1, len(marks) - 1 - skip_steps)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg, steps


def family(n):
    for key in ("conv_igemm_dma_kernel", "conv_igemm_kernel", "conv_wgrad_dma_kernel", "conv_wgrad_alltaps_kernel",
                "conv_wgrad_kernel", "wgrad_reduce_kernel", "norm_bwd_apply_kernel", "norm_bwd_reduce_kernel",
                "affine_act_bwd_kernel", "affine_act_kernel", "layernorm_act_bwd_kernel", "layernorm_act_kernel",
                "weight_prepare_t_kernel", "weight_prepare_kernel", "conv_splitk_epilogue_kernel", "conv_small_kernel",
                "conv_pyr_kernel", "conv_halo_kernel", "wgrad_reduce_small_kernel", "bn_finalize_kernel", "colsum_finalize_kernel",
                "mlp_rows_kernel", "blur_fuse_kernel", "fuse_up2_kernel"):
        if key in n:
            return key
    return None


fetch, steps = per_kernel(sys.argv[1], "FETCH_SIZE", 3)
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE", 3)
out = {}
for name in set(fetch) | set(write):
    fam = family(name)
    if fam is None:
        continue
    e = out.setdefault(fam, {"launches_per_step": 0.0, "fetch_MB_per_step_corrected": 0.0, "write_MB_per_step": 0.0})
    e["launches_per_step"] += fetch.get(name, [0, 0])[0] / steps
    e["fetch_MB_per_step_corrected"] += 2.0 * fetch.get(name, [0, 0])[1] / 1024 / steps
    e["write_MB_per_step"] += write.get(name, [0, 0])[1] / 1024 / steps
for e in out.values():
    e["traffic_MB_per_launch"] = (e["fetch_MB_per_step_corrected"] + e["write_MB_per_step"]) / max(1.0, e["launches_per_step"])
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_stamp: ties the figures to the kernel sources they were measured on)

json.dump({"steps_analysed": steps, "source_stamp": bench.source_stamp(), "note": "FETCH_SIZE doubled (gfx950 correction), KB -> MB; bench.py --no-graph, bs 32, 512^2, bf16",
           "kernels": out}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
