#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s11
mkdir -p $O
cd $R
timeout 600 python3 tools/ln_rider_time.py > $O/ln_rider_time.txt 2>&1
cat $O/ln_rider_time.txt
