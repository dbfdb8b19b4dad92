#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s3
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or streamk" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" >> $O/summary.txt
timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time.txt 2>&1; echo "time rc=$?" >> $O/summary.txt
for cfg in "64,4,2 64" "128,8,2 64" "64,4,2 8"; do set -- $cfg
  echo "== grouped launch cfg $1 pieces of $2 K tiles (default partials threshold)" >> $O/wgrad_variants.txt
  LAVT_TNG_CFG=$1 LAVT_TNG_PIECE=$2 timeout 300 python3 tools/wgrad_sk_time.py 2>&1 | grep group | sed 's/| stream-K.*//' >> $O/wgrad_variants.txt
done
timeout 1200 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pwam_gate" > $O/pytest_mod.log 2>&1; echo "mod rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_ops.log $O/pytest_mod.log; cat $O/conv_wgrad_time.txt $O/wgrad_variants.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s3/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
