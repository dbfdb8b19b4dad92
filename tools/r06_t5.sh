#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "200 4" "50 8" "72 16"; do set -- $cfg
for u in 768 1024 2048; do
  O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O
  ( export LAVT_ATTN_WPB_UNITS=$u ATTN_NWIN=$1 ATTN_HEADS=$2; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo "wpb_units=$u nwin $1 heads $2 $(python3 tools/trace_by_shape.py $T 6 10 | grep -i 'attn_bwd')"
  rm -rf $O
done; done
