#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for nw in 8 16 18 32 48; do
  O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O
  ( export ATTN_NWIN=$nw; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo nwin $nw; python3 tools/trace_by_shape.py $T 6 10 | grep -i "attn_bwd\|attn_fwd"
  rm -rf $O
done
