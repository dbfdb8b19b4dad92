#!/bin/bash
# swizzled 64-byte LDS rows in the attention kernels against the build before (.ab/liblavt_hip_base.so through LAVT_LIB_PATH): tests, kernels alone, then bench lines
cd ${GRAFT_REPO_ROOT:-.}
trace() { O=gpurun_out/r06_at; rm -rf $O; mkdir -p $O; local label=$1; shift 1
  ( export "$@"; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > /dev/null 2>&1 )
  T=$(find $O/tr -name '*kernel_trace.csv' | head -1)
  echo "$label $(python3 tools/trace_by_shape.py $T 6 10 | grep -i 'attn_bwd\|attn_fwd' | tr -s ' ' | cut -d' ' -f2,3,9-20 | tr '\n' '|')"; rm -rf $O; }
python3 -m pytest tests -m gpu -x -q -k "attn or attention or wmsa or swin_block or video" 2>&1 | tail -3
BASE=$PWD/.ab/liblavt_hip_base.so
for sh in 0 1; do for lib in base new; do
  L=""; [ $lib = base ] && L="LAVT_LIB_PATH=$BASE"
  trace "$lib shift=$sh stage2 18x16 N144" X=1 $L ATTN_SHIFT=$sh
  trace "$lib shift=$sh stage0 200x4 N144" X=1 $L ATTN_SHIFT=$sh ATTN_NWIN=200 ATTN_HEADS=4
  trace "$lib shift=$sh video 16x16 N392" X=1 $L ATTN_SHIFT=$sh ATTN_NWIN=16 ATTN_WD=8 ATTN_WS=7
  trace "$lib shift=$sh video 224x4 N392" X=1 $L ATTN_SHIFT=$sh ATTN_NWIN=224 ATTN_HEADS=4 ATTN_WD=8 ATTN_WS=7
  trace "$lib shift=$sh swin-t 72x12 N49" X=1 $L ATTN_SHIFT=$sh ATTN_NWIN=72 ATTN_HEADS=12 ATTN_WS=7
done; done
AB_OUT=r06_q_attn_swizzle_ab.txt AB_CONFIGS=$'base LAVT_LIB_PATH='$BASE$'\nswizzle X=1' bash tools/r06_ab.sh
AB_OUT=r06_q_attn_swizzle_ab_video.txt AB_ARGS="--workload video_swin_b_t8_384" AB_CONFIGS=$'base LAVT_LIB_PATH='$BASE$'\nswizzle X=1' bash tools/r06_ab.sh
