#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s18
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_modules.py -q -m gpu -k "pwam or bench_step or train_step" > $O/pytest_sub.log 2>&1; echo "sub rc=$?" >> $O/summary.txt
timeout 1200 python3 -m pytest tests/test_gpu_full.py -q -m gpu > $O/pytest_full.log 2>&1; echo "full rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 200 --no-cpu-baseline --no-profile > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/graph -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph.log 2>&1
f=$(ls $O/graph/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 400 > $O/by_shape_graph_replay.txt
rm -rf $O/graph
tail -n 3 $O/pytest_sub.log $O/pytest_full.log; cat $O/summary.txt; grep -n "pwam\|total" $O/by_shape_graph_replay.txt | head -30 | cut -c1-140
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s18/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('loss_check'))
    except Exception as e: print(f, 'ERR', e)
PY
