#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s6
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_ops.py -q -m gpu -k "wmsa_fused_forward_kernel or fp16_autocast or pwam or train_step or ddp_step or kv" > $O/pytest_sub.log 2>&1; echo "sub rc=$?" >> $O/summary.txt
timeout 1500 python3 -m pytest tests/test_gpu_full.py -q -m gpu > $O/pytest_full.log 2>&1; echo "full rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_ZERO_ARENA=0 timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench_noarena.json 2> $O/bench_noarena.err; echo "noarena rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_sub.log $O/pytest_full.log; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s6/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('loss'), d['config'].get('loss_check'))
    except Exception as e: print(f, 'ERR', e)
PY
