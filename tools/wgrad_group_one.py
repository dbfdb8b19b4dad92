#!/usr/bin/env python3
"""Runs the step's largest kernel family alone -- the grouped weight-gradient launch of a stage-2 Swin-B block at batch 2 (qkv / proj over the
2592 windowed rows, fc1 / fc2 over the 1800 tokens) -- a few times: target of the rocprofv3 --pmc passes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops
dev, bf = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(21)
M, Mw, Cc = 1800, 2592, 512
wmap = torch.randint(0, M, (Mw,), generator=g, dtype=torch.int32).to(dev)
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)
probs = [(4 * Cc, Cc, M, mk(M, 4 * Cc), mk(M, Cc), {}), (Cc, 4 * Cc, M, mk(M, Cc), mk(M, 4 * Cc), {}),
         (3 * Cc, Cc, Mw, mk(Mw, 3 * Cc), mk(M, Cc), dict(b_rowmap=wmap)), (Cc, Cc, Mw, mk(M, Cc), mk(Mw, Cc), dict(a_rowmap=wmap))]
structs, keep = [], []
class Q:
    def add(self, p, t, extra=False): structs.append(p); keep.append(t)
outs = []
for I, J, Kd, A, B, kw in probs:
    out = torch.zeros(I, J, device=dev); cs = torch.zeros(I, device=dev)
    ops.gemm_tn(bf, I, J, Kd, A, I, B, J, out, J, colsum=cs, defer=Q(), **kw)
    outs.append((out, cs))
arr = (K.GemmTN * len(structs))(*structs)
for _ in range(6):
    K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream()))
torch.cuda.synchronize()
