#!/usr/bin/env python3
"""Per-kernel breakdown of the north-star forward from a rocprofv3 --kernel-trace CSV of tools/ns_forward_trace.py.

Forwards are delimited by the probe's marker launch (a 7-element float64 fill: grid of one workgroup, the only
`FillFunctor<double>` in the trace).  Output: (1) per kernel name + grid: launches per forward, mean us, ms per forward;
(2) wall per forward, kernel-busy time, gap total; (3) with --timeline, the launch sequence of the median forward."""
import collections
import csv
import statistics
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_mark = lambda r: "FillFunctor<double>" in r["Kernel_Name"]  # noqa: E731
marks = [i for i, r in enumerate(rows) if is_mark(r)]
assert len(marks) >= 3, f"only {len(marks)} marker launches found"
fwds = [rows[a + 1:b] for a, b in zip(marks[:-1], marks[1:])]
fwds = fwds[1:]  # the first traced forward still carries warm-up effects
n = len(fwds)


def short(name):
    name = name.replace("sihl::", "")
    for k in ("void ", "(anonymous namespace)::", "at::native::"):
        name = name.replace(k, "")
    return name[:88]


def grid(r):
    g = [int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    w = [max(1, int(r.get(k, 1) or 1)) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    return "x".join(str(a // b) for a, b in zip(g, w) if a // b > 1) or "1", w[0] * w[1] * w[2]


agg = collections.defaultdict(lambda: [0, 0.0])
walls, busys = [], []
for f in fwds:
    t0, t1 = int(f[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in f)
    walls.append((t1 - t0) / 1e3)
    busy, cur_end = 0, t0
    for r in f:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
        gs, wg = grid(r)
        a = agg[(short(r["Kernel_Name"]), gs, wg)]
        a[0] += 1
        a[1] += (e - s) / 1e3
    busys.append(busy / 1e3)
print(f"forwards analysed: {n}; wall per forward {statistics.mean(walls):.1f} us (median {statistics.median(walls):.1f}); "
      f"kernel-busy {statistics.mean(busys):.1f} us; gaps {statistics.mean(walls) - statistics.mean(busys):.1f} us; "
      f"launches per forward {sum(len(f) for f in fwds) / n:.1f}")
print(f"{'us/fwd':>9} {'n/fwd':>6} {'avg us':>8}  {'workgroups':>11} {'wg':>4}  kernel")
for (name, gs, wg), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / n:9.1f} {c / n:6.1f} {us / c:8.1f}  {gs:>11} {wg:4d}  {name}")
by_name = collections.defaultdict(lambda: [0, 0.0])
for (name, gs, wg), (c, us) in agg.items():
    k = by_name[name.split("<")[0].split("(")[0]]
    k[0] += c
    k[1] += us
print("\nby kernel family:")
for name, (c, us) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / n:9.1f} us/fwd {c / n:6.1f} launches  {name}")
if "--timeline" in sys.argv:
    f = sorted(zip(walls, fwds), key=lambda p: p[0])[n // 2][1]
    t0 = int(f[0]["Start_Timestamp"])
    prev_end = t0
    print("\ntimeline of the median forward (start us, duration us, gap before us, workgroups, kernel):")
    for r in f:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gs, wg = grid(r)
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f} {gs:>9} {short(r['Kernel_Name'])[:70]}")
        prev_end = max(prev_end, e)
