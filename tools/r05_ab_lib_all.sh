#!/bin/bash
# same-box A/B of two library builds on several workloads: usage r05_ab_lib_all.sh <other.so> <workloads...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
other=$1; shift
O=$R/gpurun_out/r05_ab_lib.txt
: > $O
for w in "$@"; do
  for arm in new old new old; do
    if [ $arm = old ]; then ms=$(LAVT_LIB_PATH=$other python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    else ms=$(python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); fi
    echo "$w $arm $ms" | tee -a $O
  done
done
