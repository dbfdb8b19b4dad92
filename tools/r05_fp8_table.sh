#!/bin/bash
# graph-replay by-shape table of the fp8 batch-4 workload (and its bf16 twin) on the GPU box -> gpurun_out/r05_fp8/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_fp8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for w in swin_b_w12_480_b4_fp8 swin_b_w12_480_b4; do
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
  f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 70 > $O/r05_zz_by_shape_graph_replay_$w.txt
  rm -rf $O/graph_$w
done
