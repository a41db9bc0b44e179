#!/usr/bin/env python3
"""Per-step launch counts and time of the kernels that are NOT this library's (ATen / rocclr / MIOpen) from a rocprofv3
--kernel-trace CSV of bench.py, steady-state steps only (delimited by od_anchors_kernel as summarize_trace.py does)."""
import collections
import csv
import sys

path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "od_anchors_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-nsteps - 1], marks[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if t0 <= int(r["Start_Timestamp"]) < t1 and "anonymous namespace)::" not in r["Kernel_Name"].split("<")[0][:40] \
            or (t0 <= int(r["Start_Timestamp"]) < t1 and "at::native" in r["Kernel_Name"]):
        a = agg[r["Kernel_Name"][:150]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot_n = sum(v[0] for v in agg.values()) / nsteps
tot_t = sum(v[1] for v in agg.values()) / nsteps
print(f"non-library kernels: {tot_n:.0f} launches/step, {tot_t:.2f} ms/step")
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {c / nsteps:7.1f}/step  {ms / nsteps:7.3f} ms/step  {n[:140]}")
