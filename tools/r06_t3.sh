#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_modules.py -q -x -k "block or video or two_train or harness or train_step" 2>&1 | tail -4
AB_CONFIGS="old_attn LAVT_LIB_PATH=$PWD/lavt-rs_amd/csrc/liblavt_hip_oldattn.so
new_attn LAVT_X=1" AB_OUT=r06_e_attn_bwd_slab_ab.txt bash tools/r06_ab.sh
AB_ARGS="--workload video_swin_b_t8_384" AB_CONFIGS="old_attn LAVT_LIB_PATH=$PWD/lavt-rs_amd/csrc/liblavt_hip_oldattn.so
new_attn LAVT_X=1" AB_OUT=r06_e_attn_bwd_slab_ab_video.txt bash tools/r06_ab.sh
