#!/usr/bin/env python3
"""Correctness sweep of the v2 GEMM configurations against torch.matmul (fp32 reference of bf16 inputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops
dev, bf = "cuda:0", torch.bfloat16
torch.manual_seed(0)
for (M, N, K) in ((700, 264, 328), (256, 128, 64), (128, 128, 128), (300, 256, 512)):
    A = torch.randn(M, K, device=dev).to(bf)
    for km in (False, True):
        B = torch.randn((K, N) if km else (N, K), device=dev).to(bf)
        ref = A.float() @ (B.float() if km else B.float().t())
        for tile in ("64", "128"):
            for st in ("2", "3"):
                for wv in ("4", "8"):
                    os.environ.update(LAVT_GEMM_TILE=tile, LAVT_GEMM_STAGES=st, LAVT_GEMM_WAVES=wv)
                    C = torch.full((M, N), float("nan"), device=dev, dtype=bf)
                    ops.gemm_nt(bf, M, N, K, A, K, B, N if km else K, C, N, b_kmajor=km)
                    torch.cuda.synchronize()
                    err = (C.float() - ref).abs().max().item()
                    bad = ((C.float() - ref).abs() > 0.05 * ref.abs().max()).nonzero()
                    print(f"M{M} N{N} K{K} km={int(km)} tile{tile} s{st} w{wv}: err {err:.3f} {'OK' if err < 0.05 * ref.abs().max().item() else 'BAD n=%d first=%s' % (len(bad), bad[:3].tolist())}")
