#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; T=$O/r05_wmsa.txt; rm -f $T
python -m pytest tests -q -m gpu -k "wmsa or block or stage or full_image_config or train_step_gradients or bench_step" 2>&1 | tail -4 >> $T
echo "== previous library (2-stage ring of 64-wide K tiles)" >> $T; LAVT_LIB_PATH=$R/lavt-rs_amd/csrc/.ab/liblavt_hip_prev.so python tools/wmsa_time.py 2>/dev/null >> $T
echo "== this library (4-stage ring of 32-wide K tiles)" >> $T; python tools/wmsa_time.py 2>/dev/null >> $T
bash tools/ab_lib.sh $R/lavt-rs_amd/csrc/.ab/liblavt_hip_prev.so >> $T 2>&1
cat $T
