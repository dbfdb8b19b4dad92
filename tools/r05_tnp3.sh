#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_tnp_ablate2.txt; rm -f $T
AB=$R/lavt-rs_amd/csrc/.ab
python -m pytest tests -q -m gpu -k "tn_grouped or token_order or rider or train_step_gradients" 2>&1 | tail -3 >> $T
run() { echo "== $1" >> $T; shift; env "$@" python tools/ln_rider_time.py 2>/dev/null | head -1 >> $T; }
run "full kernel" LAVT_TN_PIPE=1
run "full kernel, XCD-contiguous tile order" LAVT_TN_PIPE=1 LAVT_PROBE=0,0,0,0,0,0,2
run "64x64 launch (gemm_tn_v2)" LAVT_TN_PIPE=0
run "no MFMA (DMA + fragment reads)" LAVT_LIB_PATH=$AB/liblavt_hip_abl1.so
run "DMA only" LAVT_LIB_PATH=$AB/liblavt_hip_abl3.so
run "no DMA in the loop (MFMA + fragment reads)" LAVT_LIB_PATH=$AB/liblavt_hip_abl4.so
run "3 stages" LAVT_TN_PIPE_STAGES=3
for v in 1 0 1 0; do LAVT_TN_PIPE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile > /tmp/b_$v.json 2>/dev/null; python - <<PY >> $T
import json
d=json.loads(open("/tmp/b_$v.json").read().strip().splitlines()[-1]); print("LAVT_TN_PIPE=$v", d["ms_per_step"], d["value"])
PY
done
cat $T
