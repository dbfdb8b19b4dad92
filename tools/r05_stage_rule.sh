#!/bin/bash
# all workloads on the library as built (the 2-stage ring from 512 workgroups up), plus LAVT_TN_PIPE_MIN_KTILES=16 on the batch-4 lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_stage_rule.txt
: > $O
run() { w=$1; shift; label=$1; shift
  ms=$(env "$@" python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
  echo "$w  $label  $ms" | tee -a $O; }
for w in swin_b_w12_480_b2 swin_b_w12_480_b4 swin_b_w12_480_b4_fp8 swin_t_w7_480_b8 video_swin_b_t8_384 video_swin_b_t8_384_sept lavt_one_swin_b_w12_480_b2; do
  run $w stages2_from_512 X=1
done
for w in swin_b_w12_480_b4 swin_b_w12_480_b4_fp8; do
  run $w MIN_KTILES=16 LAVT_TN_PIPE_MIN_KTILES=16
  run $w stages2_from_512_again X=1
done
