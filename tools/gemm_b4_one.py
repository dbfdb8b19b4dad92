#!/usr/bin/env python3
"""fc1 of a stage-2 Swin block at batch 4 (3600 x 2048 x 512, bf16) a few times: target of the rocprofv3 --pmc passes that compare the 2-stage ring (two workgroups
per CU, the default from 257 workgroups up) with the 4-stage ring (LAVT_PROBE=0,0,0,0,0,0,0,600: one workgroup per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops
dev, bf = torch.device("cuda:0"), torch.bfloat16
M, N, Kd = 3600, 2048, 512
A = torch.randn(M, Kd, device=dev).to(bf)
B = torch.randn(N, Kd, device=dev).to(bf)
C = torch.empty(M, N, device=dev, dtype=bf)
for _ in range(8):
    ops.gemm_nt(bf, M, N, Kd, A, Kd, B, Kd, C, N)
torch.cuda.synchronize()
