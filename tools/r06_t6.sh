#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn or layernorm or layer_norm" 2>&1 | tail -2
timeout 1200 python3 -m pytest tests/test_gpu_modules.py -q -x 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_gpu_full.py -q -x -k "bench_step or swin_b-bf16 or swin_t-bf16" 2>&1 | tail -2
CFG="ln_host LAVT_X=1
attn_host LAVT_DTABLE_LN_HOST=0
attn_host_unsplit LAVT_DTABLE_LN_HOST=0 LAVT_PROBE=0,0,0,0,0,0,1"
AB_CONFIGS="$CFG" AB_OUT=r06_g_dtable_riders_in_layernorm_ab.txt bash tools/r06_ab.sh
