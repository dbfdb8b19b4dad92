#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_wt.txt; rm -f $T
echo "== write-through epilogue stores (old = .ab/liblavt_hip_wt.so) vs plain (new = in-tree)" >> $T
bash tools/ab_lib.sh $R/lavt-rs_amd/csrc/.ab/liblavt_hip_wt.so >> $T 2>&1
echo "== 5-stage ring of the pipelined grouped weight gradient" >> $T
LAVT_TN_PIPE_STAGES=5 python tools/ln_rider_time.py 2>/dev/null | head -1 >> $T
for v in 5 4 5 4; do LAVT_TN_PIPE_STAGES=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stages=$v', d['ms_per_step'])" >> $T; done
cat $T
