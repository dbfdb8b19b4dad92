#!/bin/bash
# A/B bench lines on one box: each CONFIG is "label ENV=.. ENV=.."; two alternating passes
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/${AB_OUT:-r06_ab.txt}
: > $OUT
for pass in 1 2; do
  while IFS= read -r line; do
    [ -z "$line" ] && continue
    label=${line%% *}; envs=${line#* }; [ "$envs" = "$label" ] && envs=""
    ms=$(env $envs python3 bench.py --no-cpu-baseline --no-profile ${AB_ARGS} 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "pass $pass  $label  $ms" | tee -a $OUT
  done <<< "$AB_CONFIGS"
done
