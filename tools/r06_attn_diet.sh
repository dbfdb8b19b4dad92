#!/bin/bash
# attention kernels on the instruction diet (wattn_fwd_mfma_v2 / wattn_bwd_mfma_v2) against the round-5 kernels (LAVT_PROBE=0,0,0,0,0,1,1), kernels alone under
# the tracer: stage-2 headline shape (18 x 16 units of 144 tokens), stage 0, the 392-token video window (16 x 16 and 224 x 4 units), Swin-T's 49 tokens; shifted and unshifted
cd ${GRAFT_REPO_ROOT:-.}
trace() { # $1 = label, rest = env
  O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O; local label=$1; shift 1
  ( export "$@"; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo "$label $(python3 tools/trace_by_shape.py $T 6 10 | grep -i 'attn_bwd\|attn_fwd' | tr -s ' ' | cut -d' ' -f2,3,9-20 | tr '\n' '|')"
  rm -rf $O
}
python3 -m pytest tests -m gpu -x -q -k "attn or attention or wmsa or swin_block or video" 2>&1 | tail -3
for sh in 0 1; do for old in 1 0; do
  trace "old=$old shift=$sh stage2 18x16 N144" LAVT_PROBE=0,0,0,0,0,$old,$old ATTN_SHIFT=$sh
  trace "old=$old shift=$sh stage0 200x4 N144" LAVT_PROBE=0,0,0,0,0,$old,$old ATTN_SHIFT=$sh ATTN_NWIN=200 ATTN_HEADS=4
  trace "old=$old shift=$sh video 16x16 N392" LAVT_PROBE=0,0,0,0,0,$old,$old ATTN_SHIFT=$sh ATTN_NWIN=16 ATTN_WD=8 ATTN_WS=7
  trace "old=$old shift=$sh video 224x4 N392" LAVT_PROBE=0,0,0,0,0,$old,$old ATTN_SHIFT=$sh ATTN_NWIN=224 ATTN_HEADS=4 ATTN_WD=8 ATTN_WS=7
  trace "old=$old shift=$sh swin-t 72x12 N49" LAVT_PROBE=0,0,0,0,0,$old,$old ATTN_SHIFT=$sh ATTN_NWIN=72 ATTN_HEADS=12 ATTN_WS=7
done; done
