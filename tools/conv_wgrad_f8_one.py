#!/usr/bin/env python3
"""Runs the e4m3 fused-tap conv weight gradient (512->512 @120x120, batch 4: decoder conv2_2 of configs[4]) a few times: target of the rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops, _capi as K
dev, bf = torch.device("cuda:0"), torch.bfloat16
B, H, W, Cin, Cout = 4, 120, 120, 512, 512
M = B * H * W
dy = torch.randn(M, Cout, device=dev).to(bf)
x = torch.randn(M, Cin, device=dev).to(bf)
dyq, dya = ops.fp8.quantize_current(dy, "pmc-dy")
xq, xa = ops.fp8.quantize_current(x, "pmc-x")
dW = torch.zeros(Cout, Cin * 9, device=dev)
ws = int(K.lib.lavt_conv3x3_wgrad_ws(B, H, W, Cout, Cin, Cin))
scr = ops._tn_parts(ws, dev)
for _ in range(6):
    K.check(K.lib.lavt_conv3x3_wgrad_f8(K.ptr(dyq), Cout, dya, K.ptr(xq), Cin, None, 0, xa, Cin, B, H, W, Cout, Cin, K.ptr(scr), scr.numel(), K.ptr(dW), 0, ops._zero_page(dev), K.stream()))
torch.cuda.synchronize()
