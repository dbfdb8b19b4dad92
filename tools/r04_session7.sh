#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s7
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3x3" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" >> $O/summary.txt
timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time.txt 2>&1; echo "time rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile --workload swin_t_w7_480_b8 > $O/bench_swint.json 2> $O/bench_swint.err; echo "swint rc=$?" >> $O/summary.txt
timeout 1200 python3 -m pytest tests/test_gpu_full.py -q -m gpu > $O/pytest_full.log 2>&1; echo "full rc=$?" >> $O/summary.txt
tail -n 4 $O/pytest_ops.log $O/pytest_full.log; cat $O/conv_wgrad_time.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s7/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
