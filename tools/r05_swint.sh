#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_swint.txt; rm -f $T
python -m pytest tests -q -m gpu -k "pipelined_k_loop or conv3x3 or full_image_config or epilogue_column" 2>&1 | tail -4 >> $T
for v in 1 0 1 0; do LAVT_CONV_TAIL=$v python bench.py --workload swin_t_w7_480_b8 --steps 60 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAVT_CONV_TAIL=$v', d['ms_per_step'], d['value'], d['config'].get('loss_check'))" >> $T; done
cat $T
