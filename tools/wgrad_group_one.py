#!/usr/bin/env python3
"""Runs the step's largest kernel family alone -- the grouped weight-gradient launch of a stage-2 Swin-B block at batch 2 as the step issues it
since round 3 (token order: fc1 / fc2 / proj / qkv over the 1800 tokens, qkv and proj through the inverse window map, plus the column-sum side
member over the 792 padded window rows) -- a few times: target of the rocprofv3 --pmc passes (tools/pmc_passes.sh).  WINDOW_ORDER=1: the
round-2 form (qkv / proj over the 2592 windowed rows)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops, rowmaps
dev, bf = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(21)
B, H, ws, shift, Cc = 2, 30, 12, 6, 512
wmap = rowmaps.window_map(B, H, H, ws, shift, dev)
inv, pad = rowmaps.window_inverse(B, H, H, ws, shift, dev), rowmaps.window_pad_rows(B, H, H, ws, shift, dev)
M, Mw = B * H * H, wmap.numel()
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)
structs, keep = [], []
class Q:
    def add(self, p, t, extra=False): structs.append(p); keep.append(t)
def member(I, J, Kd, A, Bm, **kw):
    out = torch.zeros(I, J, device=dev); cs = kw.pop("cs", None)
    cs = torch.zeros(I, device=dev) if cs is None else cs
    ops.gemm_tn(bf, I, J, Kd, A, I, Bm, kw.pop("ldb", J), out, J, colsum=cs, defer=Q(), **kw)
    keep.append((out, cs))
    return cs
member(Cc, 4 * Cc, M, mk(M, Cc), mk(M, 4 * Cc))                    # fc2
member(4 * Cc, Cc, M, mk(M, 4 * Cc), mk(M, Cc))                    # fc1
dqkv = mk(Mw, 3 * Cc)
if os.environ.get("WINDOW_ORDER", "0") == "1":
    member(Cc, Cc, Mw, mk(M, Cc), mk(Mw, Cc), a_rowmap=wmap)
    member(3 * Cc, Cc, Mw, dqkv, mk(M, Cc), b_rowmap=wmap)
else:
    member(Cc, Cc, M, mk(M, Cc), mk(Mw, Cc), b_rowmap=inv)         # proj
    bq = member(3 * Cc, 8, pad.numel(), dqkv, ops._zero_page_tensor(dev), ldb=0, a_rowmap=pad, colsum_atomic=True, extra=True)
    member(3 * Cc, Cc, M, dqkv, mk(M, Cc), a_rowmap=inv, colsum_atomic=True, cs=bq)
arr = (K.GemmTN * len(structs))(*structs)
for _ in range(6):
    K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream()))
torch.cuda.synchronize()
print("flops_per_launch", sum(2.0 * p.I * p.J * p.K for p in structs))
