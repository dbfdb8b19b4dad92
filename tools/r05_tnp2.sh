#!/bin/bash
# round 5: what bounds the pipelined grouped weight gradient -- ablation libraries, tile-order switches, PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_tnp_ablate.txt; rm -f $T
AB=$R/lavt-rs_amd/csrc/.ab
run() { echo "== $1" >> $T; shift; env "$@" python tools/ln_rider_time.py 2>/dev/null | head -1 >> $T; }
run "full kernel" LAVT_TN_PIPE=1
run "full kernel, no XCD remap" LAVT_TN_PIPE=1 LAVT_PROBE=0,0,0,0,0,0,2
run "64x64 launch (gemm_tn_v2)" LAVT_TN_PIPE=0
run "no MFMA (DMA + fragment reads)" LAVT_LIB_PATH=$AB/liblavt_hip_abl1.so
run "DMA only" LAVT_LIB_PATH=$AB/liblavt_hip_abl3.so
run "no DMA in the loop (MFMA + fragment reads)" LAVT_LIB_PATH=$AB/liblavt_hip_abl4.so
run "3 stages" LAVT_TN_PIPE_STAGES=3
cat $T
cd /tmp && export TMPDIR=/tmp
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r05_pmc/wgrad_group_one/$c -- python3 $R/tools/wgrad_group_one.py > /dev/null 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/r05_pmc $O/r05_pmc_tnp.json > /dev/null 2>&1
python3 - <<PY
import json
d=json.load(open("$O/r05_pmc_tnp.json"))
for k,v in d.items():
    print(k, json.dumps(v.get("derived"), indent=1)); print(v["counters_per_launch"]); print(v.get("launch_us_under_profiler"))
PY
rm -rf $O/r05_pmc
