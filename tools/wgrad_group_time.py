#!/usr/bin/env python3
"""Times the stage-2 grouped weight-gradient launch (tools/wgrad_group_one.py's four problems) under the tile configuration in LAVT_TNG_CFG
("tile,waves,stages"), and checks the first and third members against a torch fp32 contraction."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops
dev, bf = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(21)
M, Mw, Cc = 1800, int(os.environ.get("MW", "2592")), 512
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
K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream()))
torch.cuda.synchronize()
ref0 = probs[0][3].float().t() @ probs[0][4].float()
ref2 = probs[2][3].float().t() @ probs[2][4].float()[wmap.long()]
e0 = float((outs[0][0] - ref0).abs().max() / ref0.abs().max())
e2 = float((outs[2][0] - ref2).abs().max() / ref2.abs().max())
c0 = float((outs[0][1] - probs[0][3].float().sum(0)).abs().max())
flops = sum(2.0 * I * J * Kd for I, J, Kd, *_ in probs)
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream()))
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for _ in range(20):
            K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream()))
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
gr.replay(); torch.cuda.synchronize()
e[0].record(); [gr.replay() for _ in range(10)]; e[1].record(); torch.cuda.synchronize()
us = e[0].elapsed_time(e[1]) * 1e3 / 200
print(f"cfg {os.environ.get('LAVT_TNG_CFG', '64,4,2'):8s}  {us:7.2f} us / launch  {flops / us * 1e-6:7.1f} TF/s  frac {flops / us * 1e-6 / 2500:.3f}   err {e0:.2e} {e2:.2e} colsum {c0:.2e}")
