"""Group a rocprofv3 kernel trace by (kernel, grid) over the last N steps: which launches carry the step time."""
import csv, sys, collections, re
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
# "auto": the number of steps in the trace = launches of a kernel that runs exactly once per step (the fused loss's finish kernel)
steps = sum("upsample_ce_finish_kernel" in r["Kernel_Name"] for r in rows) if len(sys.argv) > 2 and sys.argv[2] == "auto" else (int(sys.argv[2]) if len(sys.argv) > 2 else 10)
steps = max(steps, 1)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name).replace("__hip_bfloat16", "bf16")
    key = (name[:70], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total {tot / steps / 1e3:.2f} ms/step over {steps} steps")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{t / steps:9.1f} us/step  {n / steps:6.1f} x {t / n:7.1f} us  grid {k[1]}x{k[2]}x{k[3]} wg{k[4]}  {k[0]}")
