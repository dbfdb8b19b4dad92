#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 2600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gpu_tests.txt
tail -6 gpurun_out/r06_gpu_tests.txt
