#!/usr/bin/env python3
"""Summarise the single-counter rocprofv3 --pmc passes of tools/pmc_passes.sh: per target kernel, the average of every counter over its launches.
usage: pmc_summary.py <gpurun_out/r02_pmc> <out.json>"""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
TARGET = {"conv_one": ["gemm_nt_pipe_kernel<256, 256, false, 2, 2, 2"], "wgrad_group_one": ["gemm_tn_pipe_kernel", "gemm_tn_v2_grouped"],
          "attn_one": ["wattn_bwd_mfma", "wattn_fwd_mfma"], "wmsa_one": ["wmsa_fwd_fused_kernel"], "conv_wgrad_one": ["conv_wgrad3x3_kernel"], "conv_wgrad_f8_one": ["conv_wgrad3x3_f8_kernel"], "gemm_b4_one": ["gemm_nt_v2_kernel<128, 128"],
          "conv_fp8_one": ["gemm_nt_pipe_kernel<256, 256, false, 2, 2, 2, true"]}


def derive(c, us):
    """derived figures from the per-launch counter averages (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
    and SQ_LDS_IDX_ACTIVE cycles, summed over the chip: MI355X_MICROARCH.md)"""
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                    # per XCD
        d["kernel_cycles_per_xcd"] = round(cyc)
        d["launch_us"] = us
        d["held_clock_GHz"] = round(cyc / us * 1e-3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_array_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) > 0:          # (zero for the fp8 kernels: their MFMAs count in another class)
        d["mfma_flops"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
        mf = d["mfma_flops"] / 16384.0
        if "SQ_INSTS_VALU" in c:
            d["valu_instructions_per_mfma (SQ_INSTS_VALU / MFMA count, MFMA = flops / 16384)"] = round(c["SQ_INSTS_VALU"] / mf, 3)
        if "SQ_INSTS_SALU" in c:
            d["salu_instructions_per_mfma"] = round(c["SQ_INSTS_SALU"] / mf, 3)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        d["wave_cycles_split (of SQ_WAVE_CYCLES)"] = {k: round(c[n] / w, 3) for k, n in (
            ("parked (SQ_WAIT_ANY: s_waitcnt / barrier)", "SQ_WAIT_ANY"), ("issue stall (SQ_WAIT_INST_ANY)", "SQ_WAIT_INST_ANY"),
            ("of which LDS issue stall (SQ_WAIT_INST_LDS)", "SQ_WAIT_INST_LDS"), ("issuing (SQ_ACTIVE_INST_ANY)", "SQ_ACTIVE_INST_ANY"),
            ("VALU incl. MFMA issue (SQ_ACTIVE_INST_VALU)", "SQ_ACTIVE_INST_VALU")) if n in c}
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit_rate (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum))"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["fabric_bytes (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    return d
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
    if ent["counters_per_launch"]:
        us = ent["launch_us_under_profiler"]
        ent["derived"] = derive(ent["counters_per_launch"], us.get("GRBM_GUI_ACTIVE", sorted(us.values())[len(us) // 2]))
    if not ent["counters_per_launch"]:
        continue
    key = tgt if tgt not in res else tgt + ":" + pat          # (several patterns per target: the first one that matched a kernel keeps the plain key)
    res[key] = ent
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
