#!/usr/bin/env python3
"""LayerNorm forward / backward alone (hipGraph-timed) at the row counts / widths of the step's LayerNorms.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops, _capi as K
dev, bf = "cuda:0", torch.bfloat16
def gt(fn, n=20, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
for rows, C in ((1800, 512), (4608, 512), (7200, 256), (28800, 128), (115200, 96), (450, 1024), (18432, 256)):
    x = torch.randn(rows, C, device=dev).to(bf); dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    y = torch.empty_like(x); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev); dx = torch.empty_like(x)
    blocks = int(K.lib.lavt_layernorm_bwd_blocks(K.dt(bf), rows, C))
    ws = torch.empty(blocks * 2 * C, device=dev)
    f = lambda: K.check(K.lib.lavt_layernorm_fwd(K.dt(bf), K.ptr(x), None, K.ptr(g), K.ptr(b), K.ptr(y), K.ptr(mean), K.ptr(rstd), rows, C, 1e-5, K.stream()))
    f(); torch.cuda.synchronize()
    bw = lambda: K.check(K.lib.lavt_layernorm_bwd_partial(K.dt(bf), K.ptr(dy), K.ptr(x), None, K.ptr(g), K.ptr(mean), K.ptr(rstd), K.ptr(dx), K.ptr(ws), ws.numel(), K.ptr(dres), rows, C, K.stream()))
    tf, tb = gt(f), gt(bw)
    mb = rows * C * 2 / 1e6
    print(f"rows {rows:6d} C {C:4d} ({mb:5.1f} MB): fwd {tf:5.1f} us ({2 * mb / tf / 1e3:4.2f} TB/s)   bwd(+dres, partial) {tb:5.1f} us ({4 * mb / tb / 1e3:4.2f} TB/s)  blocks {blocks}")
