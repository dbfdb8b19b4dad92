#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s4
mkdir -p $O
cd $R
timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time.txt 2>&1; echo "time rc=$?" >> $O/summary.txt
LAVT_PROBE=0,1 timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time_noxcd.txt 2>&1; echo "time2 rc=$?" >> $O/summary.txt
timeout 300 python3 tools/wgrad_sk_time.py 2>&1 | grep group | sed 's/| stream-K.*//' > $O/wgrad_default.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-profile --workload swin_b_w12_480_b4 > $O/bench_b4.json 2> $O/bench_b4.err; echo "b4 rc=$?" >> $O/summary.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
tail -n 8 $O/pytest_all.log; cat $O/conv_wgrad_time.txt $O/conv_wgrad_time_noxcd.txt $O/wgrad_default.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s4/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('kernel','')[:80])
    except Exception as e: print(f, 'ERR', e)
PY
