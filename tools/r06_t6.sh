#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
CFG="records_off LAVT_PWAM_RECORDS=0
records_on LAVT_X=1"
AB_ARGS="--workload swin_t_w7_480_b8" AB_CONFIGS="$CFG" AB_OUT=r06_l_pwam_records_ab_swint.txt bash tools/r06_ab.sh
AB_ARGS="--workload swin_b_w12_480_b4" AB_CONFIGS="$CFG" AB_OUT=r06_l_pwam_records_ab_b4.txt bash tools/r06_ab.sh
AB_ARGS="--workload video_swin_b_t8_384" AB_CONFIGS="$CFG" AB_OUT=r06_l_pwam_records_ab_video.txt bash tools/r06_ab.sh
