#!/bin/bash
# the 128x128-tile workgroup count from which the 2-stage ring is used (LAVT_PROBE slot 7), alternating with the default (512) on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_s2min.txt
: > $O
for w in "$@"; do
  for v in 0 420 300 257 0 420 300; do
    ms=$(LAVT_PROBE=0,0,0,0,0,0,0,$v python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$w s2_min128=$v $ms" | tee -a $O
  done
done
