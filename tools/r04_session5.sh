#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s5
mkdir -p $O
cd $R
for cfg in "64,4,2" "6412,4,2" "12864,4,2"; do
  echo "== grouped launch cfg $cfg" >> $O/wgrad_variants.txt
  LAVT_TNG_CFG=$cfg timeout 300 python3 tools/wgrad_sk_time.py 2>&1 | grep group | sed 's/| stream-K.*//' >> $O/wgrad_variants.txt
done
for cfg in "6412,4,2" "12864,4,2"; do
LAVT_TNG_CFG=$cfg timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?" >> $O/summary.txt
done
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
timeout 3000 python3 -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
tail -n 12 $O/pytest_all.log; cat $O/wgrad_variants.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s5/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
