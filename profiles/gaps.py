#!/usr/bin/env python3
"""GPU idle time inside the steady-state steps of a rocprofv3 kernel trace of bench.py."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "od_anchors_kernel" in r["Kernel_Name"]]
n = 3
t0, t1 = marks[-n - 1], marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
busy_end = int(sel[0]["Start_Timestamp"])
gaps = []
busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gaps.append((s - busy_end, prev, r["Kernel_Name"][:70]))
    if e > busy_end:
        busy += e - max(s, busy_end)
        busy_end = e
    prev = r["Kernel_Name"][:70]
tot = (t1 - t0) / 1e6 / n
print(f"wall/step {tot:.2f} ms, busy/step {busy / 1e6 / n:.2f} ms, idle/step {tot - busy / 1e6 / n:.2f} ms, gaps/step {len(gaps) / n:.0f}")
big = sorted(gaps, reverse=True)[:25]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us  after [{a}]  before [{b}]")
hist = [0, 0, 0, 0]
for g, _, _ in gaps:
    hist[0 if g < 5e3 else 1 if g < 20e3 else 2 if g < 100e3 else 3] += g
print("idle ms/step by gap size (<5us, 5-20us, 20-100us, >100us):", [round(h / 1e6 / n, 2) for h in hist])
