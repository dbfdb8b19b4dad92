#!/bin/bash
# one replay of the headline step as an ordered launch list (tools/step_timeline.py); TL_TAG names the output, TL_ENV is extra environment
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06_tl
rm -rf $O; mkdir -p $O
( cd /tmp && TMPDIR=/tmp env $TL_ENV rocprofv3 --kernel-trace --output-format csv -d $O/graph -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-profile ${TL_ARGS} > $O/graph.log 2>&1 )
T=$(find $O/graph -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $T > $R/gpurun_out/r06_step_timeline_${TL_TAG:-x}.txt
python3 tools/trace_by_shape.py $T auto 60 > $R/gpurun_out/r06_by_shape_${TL_TAG:-x}.txt
head -3 $R/gpurun_out/r06_step_timeline_${TL_TAG:-x}.txt
rm -rf $O
