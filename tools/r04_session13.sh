#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s13
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_ddp1.json 2> $O/bench_ddp1.err; echo "ddp1 rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_all.log; tail -n 2 $O/smoke.log; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s13/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d['config'].get('rccl'))
    except Exception as e: print(f, 'ERR', e)
PY
