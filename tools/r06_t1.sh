#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python3 -m pytest tests/test_gpu_ops.py -q -x -k "pwam_words_moments" 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_gpu_modules.py -q -x -k "pwam" 2>&1 | tail -5
