#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1200 python3 tools/fp8_mx_emulate.py swin_b_b4 2>&1 | grep -v "^W\|Warning\|warn" | tail -12 | tee gpurun_out/r06_f_fp8_block_scaled_emulation.txt
