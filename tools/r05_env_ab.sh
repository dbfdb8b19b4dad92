#!/bin/bash
# alternate an environment setting with the default on the given workloads (one box): usage r05_env_ab.sh VAR=VALUE workloads...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
kv=$1; shift
O=$R/gpurun_out/r05_env_ab_${kv//[=,]/_}.txt
: > $O
for w in "$@"; do
  for arm in default $kv default $kv; do
    if [ $arm = default ]; then e="X=1"; else e=$kv; fi
    ms=$(env $e python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))")
    echo "$w $arm $ms" | tee -a $O
  done
done
