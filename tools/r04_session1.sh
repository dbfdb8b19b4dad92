#!/bin/bash
# round-4 GPU session 1: new parity tests, conv weight-gradient timing, stream-K timing, bench line, 1-rank-group DDP step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s1
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or token_order or fp16_autocast or forced_tiles or streamk or partial_tiles" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" >> $O/summary.txt
timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time.txt 2>&1; echo "time rc=$?" >> $O/summary.txt
for nw in 512 256 384 768; do LAVT_PROBE=$nw timeout 300 python3 tools/wgrad_sk_time.py >> $O/wgrad_sk_time.txt 2>&1; echo "sk $nw rc=$?" >> $O/summary.txt; done
timeout 1500 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "wmsa_fused_forward_kernel or pwam_gate or ddp_step" > $O/pytest_mod.log 2>&1; echo "mod rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_WGRAD_STREAMK=0 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_nosk.json 2> $O/bench_nosk.err; echo "nosk rc=$?" >> $O/summary.txt
LAVT_WGRAD_STREAMK=0 LAVT_CONV_WGRAD_TAPS=0 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_old.json 2> $O/bench_old.err; echo "old rc=$?" >> $O/summary.txt
LAVT_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_ddp1.json 2> $O/bench_ddp1.err; echo "ddp1 rc=$?" >> $O/summary.txt
tail -n 5 $O/pytest_ops.log $O/pytest_mod.log; cat $O/conv_wgrad_time.txt $O/wgrad_sk_time.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s1/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
