#!/bin/bash
# round-6 baseline on one box: bench line, PWAM node replay (total + per-launch), PMC passes of the stage-2 attention backward
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r06_a_bench_base.json 2> gpurun_out/r06_a_bench_base.err
tail -c 600 gpurun_out/r06_a_bench_base.json
O=$R/gpurun_out/r06_pw
rm -rf $O; mkdir -p $O
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/pwam_node_time.py > $O/run.log 2>&1 )
tail -2 $O/run.log
T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python3 tools/trace_by_shape.py $T 34 300 > $R/gpurun_out/r06_a_pwam_by_shape_base.txt
head -50 $R/gpurun_out/r06_a_pwam_by_shape_base.txt
rm -rf $O
PMC_ROUND=r06 bash tools/pmc_passes.sh attn_one > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/r06_pmc gpurun_out/r06_pmc_attn.json
rm -rf gpurun_out/r06_pmc
head -c 3000 gpurun_out/r06_pmc_attn.json
