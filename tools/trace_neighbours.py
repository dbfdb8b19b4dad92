#!/usr/bin/env python3
"""Which launches surround the tiny framework kernels (fills, copies) of a traced hipGraph replay: prints, for every kernel whose name matches one of the
patterns, its duration and the two launches before / after it in the last step of the trace.  usage: trace_neighbours.py <kernel_trace.csv> [pattern ...]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pats = sys.argv[2:] or ["FillFunctor", "copyBuffer", "elementwise_kernel"]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
fin = [i for i, r in enumerate(rows) if "upsample_ce_finish_kernel" in r["Kernel_Name"]]
# (the last occurrence is bench.py's eager loss check, the one before it closes the forward half of the last replay: take the replay before that, whole)
prev, last = fin[-4], fin[-3]
step = rows[prev + 1:last + 1]
for i, r in enumerate(step):
    if any(p in r["Kernel_Name"] for p in pats):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"--- [{i}/{len(step)}] {d:5.1f} us  {short(r['Kernel_Name'])}  grid {r['Grid_Size_X']}")
        for j in (i - 2, i - 1, i + 1, i + 2):
            if 0 <= j < len(step):
                print(f"      {'before' if j < i else 'after '}: {short(step[j]['Kernel_Name'])}")
