#!/usr/bin/env python3
"""Busy time, idle gaps and the per-shape kernel times of the LAST part of a rocprofv3 kernel trace (the hipGraph replays of bench.py):
python tools/trace_graph_gaps.py <kernel_trace.csv> [fraction of the trace to keep, default 0.4]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
tail = rows[int(len(rows) * (1 - frac)):]
dur = collections.defaultdict(lambda: [0, 0.0])
gap_total, busy, overlap = 0.0, 0.0, 0.0
for a, b in zip(tail, tail[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if 0 < g < 50000:
        gap_total += g
    if g < 0:
        overlap += -g
for r in tail:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = (name[:70], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    dur[k][0] += 1
    dur[k][1] += d
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print(f"kernels {len(tail)}  span {span / 1e6:.2f} ms  sum of kernel times {busy / 1e6:.2f} ms  gaps (< 50 us) {gap_total / 1e6:.2f} ms  overlap {overlap / 1e6:.2f} ms")
for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{t / busy * 100:5.1f}%  {c:5d} x {t / c / 1e3:7.1f} us  grid {k[1]}x{k[2]}x{k[3]:<4d} {k[0]}")
