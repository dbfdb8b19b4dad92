#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s10
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "layernorm_rider or token_order or streamk" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_LN_RIDER=0 timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench_norider.json 2> $O/bench_norider.err; echo "norider rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/summary.txt
timeout 2400 python3 -m pytest tests/test_gpu_modules.py tests/test_gpu_full.py -q -m gpu > $O/pytest_mod.log 2>&1; echo "mod rc=$?" >> $O/summary.txt
tail -n 6 $O/pytest_ops.log $O/pytest_mod.log; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s10/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('loss_check'))
    except Exception as e: print(f, 'ERR', e)
PY
