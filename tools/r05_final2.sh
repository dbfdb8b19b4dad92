#!/bin/bash
# final validation + evidence of the second session of round 5: the whole GPU test-suite, the smoke entry, bench lines of every workload, graph-replay tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r05_final2
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
cd $R
LAVT_PROFILE_DUMP=$O/r05_zzz_families_by_scope.txt python3 bench.py > $O/r05_zzz_bench_line.json 2> $O/bench.err
for w in swin_t_w7_480_b8 video_swin_b_t8_384 video_swin_b_t8_384_sept swin_b_w12_480_b4 swin_b_w12_480_b4_fp8 lavt_one_swin_b_w12_480_b2; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 100 > $O/r05_zzz_bench_line_$w.json 2>> $O/bench.err
done
rocprofv3 --kernel-trace --output-format csv -d $O/graph -- python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph.log 2>&1
f=$(ls $O/graph/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 400 > $O/r05_zzz_by_shape_graph_replay.txt
python3 tools/step_timeline.py $f > $O/r05_zzz_step_timeline.txt
rm -rf $O/graph
for w in swin_t_w7_480_b8 video_swin_b_t8_384 swin_b_w12_480_b4 swin_b_w12_480_b4_fp8; do
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
  f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 70 > $O/r05_zzz_by_shape_graph_replay_$w.txt
  rm -rf $O/graph_$w
done
for f in $O/r05_zzz_bench_line*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['unit'], d.get('roofline',{}).get('frac'))"; done
