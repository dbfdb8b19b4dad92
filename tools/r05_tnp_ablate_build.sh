#!/bin/bash
# ablation libraries of csrc/gemm_tn_pipe.hip (TNP_ABL bits: 1 no MFMA, 2 no fragment reads, 4 no DMA in the K loop) under lavt-rs_amd/csrc/.ab/ (git-ignored)
cd "$(dirname "$0")/../lavt-rs_amd/csrc" || exit 1
mkdir -p .ab
OBJS=$(ls *.o | grep -v gemm_tn_pipe.o)
for a in ${@:-1 3 4}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DTNP_ABL=$a -c gemm_tn_pipe.hip -o .ab/gemm_tn_pipe_$a.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o .ab/liblavt_hip_abl$a.so $OBJS .ab/gemm_tn_pipe_$a.o && rm .ab/gemm_tn_pipe_$a.o
done
ls -la .ab
