#!/bin/bash
# Evidence run on the GPU box (gpurun; RND=r04 by default): bench lines of every BASELINE workload, the in-process family table, and rocprofv3 kernel traces
# (eager: --stats summary as in rounds 1-2; hipGraph replay: grouped by kernel and grid).  Writes under gpurun_out/${RND}_profiles/.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${RND:-r04}
O=$R/gpurun_out/${RND}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
LAVT_PROFILE_DUMP=$O/${RND}_z_families_by_scope.txt python3 bench.py > $O/${RND}_z_bench_line.json 2> $O/bench.err
for w in swin_t_w7_480_b8 video_swin_b_t8_384 video_swin_b_t8_384_sept swin_b_w12_480_b4 swin_b_w12_480_b4_fp8 lavt_one_swin_b_w12_480_b2; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 100 > $O/${RND}_z_bench_line_$w.json 2>> $O/bench.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- python3 bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-profile --no-optimizer > $O/eager.log 2>&1
f=$(ls $O/eager/*/*kernel_stats.csv | head -1); cp $f $O/${RND}_z_kernel_stats_bf16_swinb_b2.csv
f=$(ls $O/eager/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 80 > $O/${RND}_z_by_shape.txt
rm -rf $O/eager
rocprofv3 --kernel-trace --output-format csv -d $O/graph -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph.log 2>&1
f=$(ls $O/graph/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 400 > $O/${RND}_z_by_shape_graph_replay.txt
python3 tools/step_timeline.py $f > $O/${RND}_z_step_timeline.txt          # one replay as an ordered launch list
rm -rf $O/graph
# by-shape tables (hipGraph replay) of the non-headline workloads, next to their bench lines
for w in swin_t_w7_480_b8 video_swin_b_t8_384 swin_b_w12_480_b4 swin_b_w12_480_b4_fp8; do
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
  f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 60 > $O/${RND}_z_by_shape_graph_replay_$w.txt
  rm -rf $O/graph_$w
done
ls -la $O
