#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python3 -m pytest tests/test_gpu_ops.py -q -x -k "pwam" 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_modules.py -q -x -k "pwam or harness or train_step" 2>&1 | tail -3
for t in ${TAILS:-0 1}; do
  O=gpurun_out/r06_pw_$t; rm -rf $O; mkdir -p $O
  ( export LAVT_PWAM_RECORDS=$t; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/pwam_node_time.py > $GRAFT_REPO_ROOT/$O/run.log 2>&1 )
  tail -1 $O/run.log
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_by_shape.py $T 34 300 | grep -i "pwam\|total\|tn_reduce\|colstats\|gemm_tn_v2_kernel<64, 64, 4, 2, false, 1" > gpurun_out/r06_b_pwam_by_shape_records_$t.txt
  rm -rf $O
done
grep -i "total\|mix\|bwd1" gpurun_out/r06_b_pwam_by_shape_records_*.txt | cut -c1-190
