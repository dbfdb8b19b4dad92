#!/bin/bash
# experiment library: the NT GEMM families with PLAIN epilogue stores (-DLAVT_ST_PLAIN; the shipped build stores write-through) under lavt-rs_amd/csrc/.ab/liblavt_hip_plainst.so
cd "$(dirname "$0")/../lavt-rs_amd/csrc" || exit 1
mkdir -p .ab
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DLAVT_ST_PLAIN=1"
/opt/rocm/bin/hipcc $F -c gemm_v2.hip -o .ab/gemm_v2_plainst.o &
/opt/rocm/bin/hipcc $F -c gemm_nt_pipe.hip -o .ab/gemm_nt_pipe_plainst.o &
wait
OBJS=$(ls *.o | grep -v "^gemm_v2.o\|^gemm_nt_pipe.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o .ab/liblavt_hip_plainst.so $OBJS .ab/gemm_v2_plainst.o .ab/gemm_nt_pipe_plainst.o && rm .ab/*_plainst.o
ls -la .ab/liblavt_hip_plainst.so
