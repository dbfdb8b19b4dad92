#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_ln.txt; rm -f $T
python -m pytest tests -q -m gpu -k "layernorm or layer_norm or rider or block or train_step_gradients or tn_grouped" 2>&1 | tail -3 >> $T
python tools/ln_bench.py 2>/dev/null >> $T
for w in swin_b_w12_480_b2 swin_t_w7_480_b8; do python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['config']['loss_check']['ok'])" >> $T; done
cat $T
