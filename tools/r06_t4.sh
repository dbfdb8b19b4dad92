#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for lib in liblavt_hip_oldattn.so liblavt_hip.so; do
  O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O
  ( export LAVT_LIB_PATH=$GRAFT_REPO_ROOT/lavt-rs_amd/csrc/$lib; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo $lib; python3 tools/trace_by_shape.py $T 6 10 | grep -i "attn\|dtable"
  rm -rf $O
done
