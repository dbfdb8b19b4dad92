#!/usr/bin/env python3
"""Runs the bench's dominant kernel (bf16 implicit-GEMM 3x3 conv, 512->512 @120x120, batch 2) a few times: target of the rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops
B, H, W, Cin, Cout = 2, 120, 120, 512, 512
x = torch.randn(B * H * W, Cin, device="cuda:0").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda:0") * (9 * Cin) ** -0.5
with torch.no_grad():
    for _ in range(6):
        y = ops.conv3x3(x, None, w, B, H, W)
torch.cuda.synchronize()
