#!/bin/bash
# end-of-round-4 evidence of the fp8 path (configs[4]): graph-replay by-shape tables of Swin-B batch 4 with and without fp8, the conv A/B
# (pipelined fp8 kernel / gemm_v2 fp8 K loop / bf16) and the PMC passes of the pipelined fp8 convolution.  Writes under gpurun_out/r04_fp8_profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_fp8_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 tools/fp8_conv_bench.py 2>/dev/null > $O/r04_fp8_conv_bench.txt
LAVT_FP8_PIPE=0 python3 tools/fp8_conv_bench.py 2>/dev/null > $O/r04_fp8_conv_bench_gemm_v2_loop.txt
for w in swin_b_w12_480_b4 swin_b_w12_480_b4_fp8; do
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-optimizer > $O/graph_$w.log 2>&1
  f=$(ls $O/graph_$w/*/*kernel_trace.csv | head -1); python3 tools/trace_by_shape.py $f auto 60 > $O/r04_z_by_shape_graph_replay_$w.txt
  rm -rf $O/graph_$w
done
PMC_ROUND=r04fp8 bash tools/pmc_passes.sh conv_fp8_one > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $R/gpurun_out/r04fp8_pmc $O/r04_pmc_conv_fp8.json >> $O/pmc.log 2>&1
rm -rf $R/gpurun_out/r04fp8_pmc
ls -la $O
