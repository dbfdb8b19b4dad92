#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_side.txt; rm -f $T
python -m pytest tests -q -m gpu -k "linear or gate or pwam or mlp or block or stage or train_step or residual" 2>&1 | tail -4 >> $T
for v in 1 0 1 0; do LAVT_SIDE_PRE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAVT_SIDE_PRE=$v', d['ms_per_step'], d['value'], d['config']['loss_check']['ok'])" >> $T; done
cat $T
