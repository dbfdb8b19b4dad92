#!/bin/bash
# one-knob-at-a-time sweep of the library's dispatch switches on the non-headline workloads (GPU box): is any default wrong for batch 4 / fp8 / Swin-T / video?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_knobs.txt
: > $O
run() {   # workload, label, env...
  w=$1; shift; label=$1; shift
  ms=$(env "$@" python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
  echo "$w  $label  $ms" | tee -a $O
}
for w in ${WORKLOADS:-swin_b_w12_480_b4_fp8 swin_t_w7_480_b8 video_swin_b_t8_384 swin_b_w12_480_b2}; do
  run $w default X=1
  run $w GEMM_PIPE=3 LAVT_GEMM_PIPE=3
  run $w GEMM_BIG_LONG=64 LAVT_GEMM_BIG_LONG=64
  run $w GEMM_BIG_LONG=256 LAVT_GEMM_BIG_LONG=256
  run $w TN_PIPE=1 LAVT_TN_PIPE=1
  run $w TN_PIPE_MIN_KTILES=8 LAVT_TN_PIPE_MIN_KTILES=8
  run $w TN_PIPE_MIN_KTILES=20 LAVT_TN_PIPE_MIN_KTILES=20
  run $w TN_PIPE_MIN_TILES=64 LAVT_TN_PIPE_MIN_TILES=64
  run $w TN_PIPE_STAGES=3 LAVT_TN_PIPE_STAGES=3
  run $w TNG_CHAIN=64 LAVT_TNG_CHAIN=64
  run $w TN_TARGET=512 LAVT_TN_TARGET=512
  run $w TN_TARGET=1024 LAVT_TN_TARGET=1024
  run $w GEMM_STAGES=2 LAVT_GEMM_STAGES=2
  run $w GEMM_STAGES=4 LAVT_GEMM_STAGES=4
  run $w default_again X=1
done
