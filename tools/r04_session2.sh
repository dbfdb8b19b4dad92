#!/bin/bash
# round-4 GPU session 2: pipelined conv wgrad, 128x128 grouped-wgrad variants, PMC of the conv wgrad kernel, bench + 1-rank DDP
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s2
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or token_order or streamk or partial_tiles" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" >> $O/summary.txt
timeout 600 python3 tools/conv_wgrad_time.py > $O/conv_wgrad_time.txt 2>&1; echo "time rc=$?" >> $O/summary.txt
echo "== stream-K (128-register cap), runs 512 / 256" >> $O/wgrad_variants.txt
for nw in 512 256; do LAVT_PROBE=$nw timeout 300 python3 tools/wgrad_sk_time.py 2>&1 | grep group >> $O/wgrad_variants.txt; done
for cfg in "128,8,2 8" "128,8,2 10" "128,8,2 15" "128,8,2 64" "64,4,2 15" "64,4,2 10"; do set -- $cfg
  echo "== grouped launch cfg $1 pieces of $2 K tiles through partial tiles" >> $O/wgrad_variants.txt
  LAVT_WGRAD_STREAMK_OFF=1 LAVT_TN_PARTIALS_MINK=512 LAVT_TNG_CFG=$1 LAVT_TNG_CHAIN=1 LAVT_TNG_PIECE=$2 timeout 300 python3 tools/wgrad_sk_time.py 2>&1 | grep group | sed 's/| stream-K.*//' >> $O/wgrad_variants.txt
done
timeout 1200 python3 -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "wmsa_fused_forward_kernel or pwam_gate or ddp_step" > $O/pytest_mod.log 2>&1; echo "mod rc=$?" >> $O/summary.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
LAVT_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile > $O/bench_ddp1.json 2> $O/bench_ddp1.err; echo "ddp1 rc=$?" >> $O/summary.txt
LAVT_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-profile --bf16-buckets > $O/bench_ddp1_bf16.json 2> $O/bench_ddp1_bf16.err; echo "ddp1bf16 rc=$?" >> $O/summary.txt
PMC_ROUND=r04 timeout 1500 bash tools/pmc_passes.sh conv_wgrad_one > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $R/gpurun_out/r04_pmc $O/r04_pmc_conv_wgrad.json >> $O/pmc.log 2>&1
rm -rf $R/gpurun_out/r04_pmc
tail -n 5 $O/pytest_ops.log $O/pytest_mod.log; cat $O/conv_wgrad_time.txt $O/wgrad_variants.txt; cat $O/summary.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_s2/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d['config'].get('profile',{}).get('wmsa_pwam_mfma_frac'))
    except Exception as e: print(f, 'ERR', e)
PY
python3 -c "
import json; d=json.load(open('$O/r04_pmc_conv_wgrad.json')); print(json.dumps(d.get('conv_wgrad_one',{}).get('derived'),indent=1))"
