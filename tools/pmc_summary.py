#!/usr/bin/env python3
"""Summarise the single-counter rocprofv3 --pmc passes of tools/pmc_passes.sh: per target kernel, the average of every counter over its launches.
usage: pmc_summary.py <gpurun_out/r02_pmc> <out.json>"""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
TARGET = {"conv_one": ["gemm_nt_v2_kernel<256, 256, false, 2, 16, 2"], "wgrad_group_one": ["gemm_tn_v2_grouped_kernel"],
          "attn_one": ["wattn_bwd_mfma", "wattn_fwd_mfma"]}
res = {}
for tgt, pat in [(t, p) for t, ps in TARGET.items() for p in ps]:
    if not os.path.isdir(os.path.join(root, tgt)):
        continue
    ent = {"kernel_match": pat, "counters_per_launch": {}}
    for d in sorted(glob.glob(os.path.join(root, tgt, "*"))):
        name = os.path.basename(d)
        fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if not fs:
            continue
        vals, durs, regs = [], [], None
        for r in csv.DictReader(open(fs[0])):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                regs = (int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"]), int(r["Scratch_Size"]), int(r["LDS_Block_Size"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        if vals:
            vals, durs = vals[1:] or vals, durs[1:] or durs          # drop the first (cold) launch
            ent["counters_per_launch"][name] = sum(vals) / len(vals)
            ent.setdefault("launch_us_under_profiler", {})[name] = round(sum(durs) / len(durs), 1)
            ent["vgpr_agpr_sgpr_scratch_lds_grid_wg"] = regs
    res[tgt if len(TARGET[tgt]) == 1 else tgt + ":" + pat] = ent
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
