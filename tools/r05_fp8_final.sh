#!/bin/bash
# second-session evidence of round 5 (fp8 weight gradient + twins): bench lines of batch 4 in bf16 / fp8 and of the headline, graph-replay tables -> gpurun_out/r05_fp8/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_fp8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py --no-cpu-baseline > $O/r05_zz_bench_line.json 2> $O/bench.err
for w in swin_b_w12_480_b4 swin_b_w12_480_b4_fp8; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 100 > $O/r05_zz_bench_line_$w.json 2>> $O/bench.err
done
for w in swin_b_w12_480_b4_fp8 swin_b_w12_480_b4; do
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
  f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 70 > $O/r05_zz_by_shape_graph_replay_$w.txt
  rm -rf $O/graph_$w
done
LAVT_FP8_WGRAD=0 LAVT_FP8_TWINS=0 python3 bench.py --workload swin_b_w12_480_b4_fp8 --no-cpu-baseline --no-profile --steps 100 > $O/r05_zz_bench_line_swin_b_w12_480_b4_fp8_wgrad_twins_off.json 2>> $O/bench.err
LAVT_FP8_TWINS=0 python3 bench.py --workload swin_b_w12_480_b4_fp8 --no-cpu-baseline --no-profile --steps 100 > $O/r05_zz_bench_line_swin_b_w12_480_b4_fp8_twins_off.json 2>> $O/bench.err
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r05_fp8/r05_zz_bench_line*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["value"], d["config"].get("loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
