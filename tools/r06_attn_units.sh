#!/bin/bash
# the attention kernels by unit count (profiles/r06_e_attn_bwd_ds_from_slab_negative.txt): stage-2 shape, 16 heads x ATTN_NWIN windows of 144 tokens,
# backward with and without the surplus-unit split (LAVT_ATTN_BWD_SPLIT), then the fused W-MSA forward by windows (tools/wmsa_one.py: WMSA_B images of WMSA_H x WMSA_H tokens)
cd ${GRAFT_REPO_ROOT:-.}
trace() { # $1 = label, $2 = script, rest = env
  O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O; local label=$1 script=$2; shift 2
  ( export "$@"; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/$script > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo "$label $(python3 tools/trace_by_shape.py $T 6 10 | grep -i 'attn_bwd\|attn_fwd\|wmsa_fwd' | tr -s ' ' | tr '\n' '|')"
  rm -rf $O
}
for nw in 8 16 18 20 24 32 36 48; do for sp in 1 0; do trace "split=$sp nwin=$nw" attn_one.py LAVT_ATTN_BWD_SPLIT=$sp ATTN_NWIN=$nw; done; done
for cfg in "1 30" "4 24" "2 30" "5 24" "6 24" "8 24"; do set -- $cfg; trace "fused forward B=$1 H=$2" wmsa_one.py WMSA_B=$1 WMSA_H=$2; done
