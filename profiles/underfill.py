#!/usr/bin/env python3
"""How much of a steady-state step is spent in launches that cannot fill the 256 CUs (rocprofv3 --kernel-trace CSV of
bench.py): per kernel name, time in launches with < 256 workgroups and with 256..1023, and the idle time."""
import collections
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "od_anchors_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-n - 1], marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
small = collections.defaultdict(lambda: [0, 0.0])
mid = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in sel:
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    nwg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    key = r["Kernel_Name"][:90]
    if nwg < 256:
        small[key][0] += 1
        small[key][1] += d
    elif nwg < 1024:
        mid[key][0] += 1
        mid[key][1] += d
queues = collections.Counter(r.get("Queue_Id", "?") for r in sel)
print(f"wall/step {(t1 - t0) / 1e6 / n:.2f} ms; kernel time/step {tot / n:.2f} ms; launches/step {len(sel) / n:.0f}; queues {dict(queues)}")
for name, d in (("< 256 workgroups", small), ("256..1023 workgroups", mid)):
    s = sum(v[1] for v in d.values()) / n
    c = sum(v[0] for v in d.values()) / n
    print(f"{name}: {s:.2f} ms/step in {c:.0f} launches/step")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"   {v[1] / n:7.3f} ms/step {v[0] / n:6.1f}/step avg {v[1] / v[0] * 1e3:6.1f} us  {k}")
