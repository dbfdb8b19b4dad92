#!/bin/bash
# one replay of the headline step as an ordered launch list (tools/step_timeline.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_tl
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/graph -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-profile > $O/graph.log 2>&1
T=$(find $O/graph -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $T > $R/gpurun_out/r05_step_timeline.txt
head -3 $R/gpurun_out/r05_step_timeline.txt
rm -rf $O
