#!/usr/bin/env python3
"""The three C x C GEMMs behind PWAM's mix kernel at stage 0 (28 800 x 128 x 128: project_mm with GELU + pre-activation, gate 1 with ReLU, gate 2 with tanh + residual + multiplier)
as the step issues them: run under rocprofv3 --kernel-trace with LAVT_GEMM_TILE unset / 64 to compare tile configurations on a K = 128 problem (two K tiles: all prologue and epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops
dev, bf = "cuda:0", torch.bfloat16
M, C = int(os.environ.get("PG_M", 28800)), int(os.environ.get("PG_C", 128))
mk = lambda r, c: (torch.randn(r, c) * 0.5).to(dev).to(bf)
mm, W, x, r = mk(M, C), mk(C, C), mk(M, C), mk(M, C)
bias = torch.randn(C, device=dev)
o1, p1, o2, o3, p3 = (torch.empty(M, C, dtype=bf, device=dev) for _ in range(5))
for _ in range(6):
    ops.gemm_nt(bf, M, C, C, mm, C, W, C, o1, C, bias=bias, act=K.ACT_GELU, Cpre=p1, ldcpre=C)
    ops.gemm_nt(bf, M, C, C, o1, C, W, C, o2, C, act=K.ACT_RELU)
    ops.gemm_nt(bf, M, C, C, o2, C, W, C, o3, C, act=K.ACT_TANH, Cpre=p3, ldcpre=C, R=x, ldr=C, mul=r, ldmul=C)
torch.cuda.synchronize()
