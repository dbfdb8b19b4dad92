#!/usr/bin/env python3
"""Runs the e4m3 implicit-GEMM 3x3 conv (512->512 @120x120, batch 2: the pipelined fp8 kernel, gemm_nt_pipe_kernel<256, 256, false, 2, 2, 2, true>) a few
times: target of the rocprofv3 --pmc passes (tools/pmc_passes.sh conv_fp8_one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops
B, H, W, Cin, Cout = 2, 120, 120, 512, 512
x = torch.randn(B * H * W, Cin, device="cuda:0").to(torch.bfloat16)
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device="cuda:0") * (9 * Cin) ** -0.5)
with lavt_hip.use_dtype("fp8"), torch.no_grad():
    for _ in range(6):
        ops.fp8.advance()
        y = ops.conv3x3(x, None, w, B, H, W)
torch.cuda.synchronize()
