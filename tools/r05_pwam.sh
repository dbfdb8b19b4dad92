#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_pw
rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_gpu_modules.py -q -k "pwam" 2>&1 | tail -3
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/pwam_node_time.py > $O/run.log 2>&1
tail -2 $O/run.log
T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python3 tools/trace_by_shape.py $T 34 300 | grep -i "pwam\|total" > $R/gpurun_out/r05_pwam_by_shape.txt
cat $R/gpurun_out/r05_pwam_by_shape.txt
rm -rf $O
