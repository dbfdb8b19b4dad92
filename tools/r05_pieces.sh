#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_pieces.txt; rm -f $T
python -m pytest tests -q -m gpu -k "tn_grouped or token_order or rider or partial_tiles or train_step_gradients or bench_step" 2>&1 | tail -4 >> $T
for v in 2 1 0; do echo "LAVT_TN_PIPE=$v" >> $T; LAVT_TN_PIPE=$v python tools/ln_rider_time.py 2>/dev/null >> $T; done
for v in 2 1 2 1; do LAVT_TN_PIPE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAVT_TN_PIPE=$v', d['ms_per_step'], d['value'], d['config']['loss_check']['ok'])" >> $T; done
for v in 2 1; do LAVT_TN_PIPE=$v python bench.py --workload swin_t_w7_480_b8 --steps 60 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin_t LAVT_TN_PIPE=$v', d['ms_per_step'], d['config']['loss_check']['ok'])" >> $T; done
cat $T
