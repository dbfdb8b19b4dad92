#!/usr/bin/env python3
"""One hipGraph replay of the training step as an ordered list of launches: offset from the step's first launch, duration, the idle gap in front of the
launch (start - previous end), grid, kernel.  Read from a `rocprofv3 --kernel-trace --output-format csv` trace of `bench.py --steps N` (a replay is
delimited by `upsample_ce_finish_kernel`, the loss: the launches after it are the backward of the same step).
usage: step_timeline.py <kernel_trace.csv> [> profiles/rNN_step_timeline.txt]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:110]
fin = [i for i, r in enumerate(rows) if "upsample_ce_finish_kernel" in r["Kernel_Name"]]
# the last occurrence is bench.py's eager loss check.  Replays are separated by the host's replay call: cut the trace at every idle gap > 6 us in front of
# a launch and take the last piece that holds exactly one loss and at least 100 launches.
cuts = [0] + [i + 1 for i in range(len(rows) - 1) if int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) > 6000] + [len(rows)]
pieces = [(cuts[j], cuts[j + 1]) for j in range(len(cuts) - 1)]
good = [(a, b) for a, b in pieces if b - a >= 100 and sum(a <= f < b for f in fin) == 1 and b <= fin[-1]]
first, last = good[-1]
step = rows[first:last]
t0 = int(step[0]["Start_Timestamp"])
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
print(f"# {len(step)} launches, span {span:.1f} us, sum of kernel durations {busy:.1f} us, idle between launches {span - busy:.1f} us")
print(f"# {'#':>4s} {'at us':>8s} {'dur us':>7s} {'gap us':>6s}  {'grid':>12s} wg    kernel")
prev_end = t0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = f"{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)}x{int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1)}"
    print(f"  {i:4d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f}  {grid:>12s} {r['Workgroup_Size_X']:>4s}  {short(r['Kernel_Name'])}")
    prev_end = e
