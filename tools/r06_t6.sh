#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -x -k "layernorm or layer_norm or attention or attn or reduce" 2>&1 | tail -2
timeout 1200 python3 -m pytest tests/test_gpu_modules.py -q -x -k "harness or train_step or block" 2>&1 | tail -2
TL_TAG=lnhost3 bash tools/r06_timeline.sh
tail -4 gpurun_out/r06_step_timeline_lnhost3.txt | cut -c1-150
