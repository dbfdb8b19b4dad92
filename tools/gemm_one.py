#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py kind M N K [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops
kind, a, b, c = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev, bf = "cuda:0", torch.bfloat16
if kind in ("nt", "ntk"):
    km = kind == "ntk"
    A = torch.randn(a, c, device=dev).to(bf); B = torch.randn((c, b) if km else (b, c), device=dev).to(bf); C = torch.empty(a, b, device=dev, dtype=bf)
    fn = lambda: ops.gemm_nt(bf, a, b, c, A, c, B, b if km else c, C, b, b_kmajor=km)
else:
    A = torch.randn(c, a, device=dev).to(bf); B = torch.randn(c, b, device=dev).to(bf); C = torch.zeros(a, b, device=dev)
    fn = lambda: ops.gemm_tn(bf, a, b, c, A, a, B, b, C, b)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
