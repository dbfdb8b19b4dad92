#!/bin/bash
# alternate LAVT_PROBE slot 7 = $1 with the default on the given workloads (one box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
v=$1; shift
O=$R/gpurun_out/r05_probe7_$v.txt
: > $O
for w in "$@"; do
  for arm in 0 $v 0 $v; do
    ms=$(LAVT_PROBE=0,0,0,0,0,0,0,$arm python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))")
    echo "$w probe7=$arm $ms" | tee -a $O
  done
done
