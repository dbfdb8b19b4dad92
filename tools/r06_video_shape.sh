#!/bin/bash
# by-shape table + ordered launch list of one hipGraph replay of the video workload -> gpurun_out/r06_profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
w=${1:-video_swin_b_t8_384}
rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 80 > $O/r06_zz_by_shape_graph_replay_$w.txt
python3 tools/step_timeline.py $f > $O/r06_zz_step_timeline_$w.txt
rm -rf $O/graph_$w
head -50 $O/r06_zz_by_shape_graph_replay_$w.txt | cut -c1-170
