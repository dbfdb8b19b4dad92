#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_tnp_ablate3.txt; rm -f $T
AB=$R/lavt-rs_amd/csrc/.ab
echo "== one K tile per output tile (fixed cost of the launch)" >> $T; LAVT_LIB_PATH=$AB/liblavt_hip_abl8.so python tools/ln_rider_time.py 2>/dev/null | head -1 >> $T
echo "== K scaling, pipelined 128x128 launch" >> $T; TNP_SCALING=1 LAVT_TN_PIPE=1 python tools/ln_rider_time.py 2>/dev/null >> $T
echo "== K scaling, 64x64 launch" >> $T; TNP_SCALING=1 LAVT_TN_PIPE=0 python tools/ln_rider_time.py 2>/dev/null >> $T
cat $T
