#!/bin/bash
# second one-knob-at-a-time sweep: the host-side thresholds (lavt_hip/ops.py) and the remaining library switches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_knobs2.txt
: > $O
run() { w=$1; shift; label=$1; shift
  ms=$(env "$@" python3 bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-optimizer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
  echo "$w  $label  $ms" | tee -a $O; }
for w in ${WORKLOADS:-swin_b_w12_480_b4 swin_t_w7_480_b8 video_swin_b_t8_384 swin_b_w12_480_b2}; do
  run $w default X=1
  run $w ATTN_BWD_WAVES=4 LAVT_ATTN_BWD_WAVES=4
  run $w CONV_SPLIT_ROWS=4096 LAVT_CONV_SPLIT_ROWS=4096
  run $w CONV_SPLIT_ROWS=1024 LAVT_CONV_SPLIT_ROWS=1024
  run $w CONV_SPLIT_N=2 LAVT_CONV_SPLIT_N=2
  run $w CONV_SPLIT_N=4 LAVT_CONV_SPLIT_N=4
  run $w TN_PARTIALS_MINK=1024 LAVT_TN_PARTIALS_MINK=1024
  run $w TN_PARTIALS_MINK=4096 LAVT_TN_PARTIALS_MINK=4096
  run $w TN_BIG_MIN=32 LAVT_TN_BIG_MIN=32
  run $w TN_BIG_MIN=96 LAVT_TN_BIG_MIN=96
  run $w TNG_PIECE=4 LAVT_TNG_PIECE=4
  run $w TNG_PIECE=16 LAVT_TNG_PIECE=16
  run $w GEMM_EPI=l LAVT_GEMM_EPI=l
  run $w LN_BWD_WAVES=8 LAVT_LN_BWD_WAVES=8
  run $w FILL_RIDERS=1 LAVT_FILL_RIDERS=1
  run $w WMSA_FUSED_MAX_C=512 LAVT_WMSA_FUSED_MAX_C=512
  run $w default_again X=1
done
run swin_b_w12_480_b4_fp8 default X=1
run swin_b_w12_480_b4_fp8 FP8_CONV_MIN_TILES=100 LAVT_FP8_CONV_MIN_TILES=100
run swin_b_w12_480_b4_fp8 FP8_CONV_MIN_TILES=50 LAVT_FP8_CONV_MIN_TILES=50
run swin_b_w12_480_b4_fp8 FP8_CONV_MIN_TILES=500 LAVT_FP8_CONV_MIN_TILES=500
run swin_b_w12_480_b4_fp8 default_again X=1
