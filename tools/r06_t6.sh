#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
CFG="wpb768 LAVT_ATTN_WPB_UNITS=768
wpb2048 LAVT_X=1"
AB_CONFIGS="$CFG" AB_OUT=r06_j_attn_wpb_ab.txt bash tools/r06_ab.sh
AB_ARGS="--workload swin_t_w7_480_b8" AB_CONFIGS="$CFG" AB_OUT=r06_j_attn_wpb_ab_swint.txt bash tools/r06_ab.sh
AB_ARGS="--workload swin_b_w12_480_b4" AB_CONFIGS="$CFG" AB_OUT=r06_j_attn_wpb_ab_b4.txt bash tools/r06_ab.sh
AB_ARGS="--workload video_swin_b_t8_384" AB_CONFIGS="$CFG" AB_OUT=r06_j_attn_wpb_ab_video.txt bash tools/r06_ab.sh
