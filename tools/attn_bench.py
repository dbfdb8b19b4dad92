#!/usr/bin/env python3
"""hipGraph-timed window-attention kernels at the Swin-B w12 / batch-2 stage shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops, rowmaps
dev, bf = "cuda:0", torch.bfloat16

def timeit(fn, iters=10, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3

for name, B, H, heads in (("s0", 2, 120, 4), ("s1", 2, 60, 8), ("s2", 2, 30, 16), ("s3", 2, 15, 32)):
    ws, C = 12, heads * 32
    Hp = rowmaps.padded(H, ws); nW = (Hp // ws) ** 2; N = ws * ws
    qkv = torch.randn(B * nW * N, 3 * C, device=dev).to(bf).requires_grad_(True)
    table = (torch.randn((2 * ws - 1) ** 2, heads, device=dev) * 0.5).requires_grad_(True)
    region = rowmaps.region_ids(H, H, ws, ws // 2, dev)
    out = ops.window_attention(qkv, table, region, ws, heads)
    go = torch.randn_like(out)
    tf = timeit(lambda: ops.window_attention(qkv, table, region, ws, heads))
    def fb():
        o = ops.window_attention(qkv, table, region, ws, heads)
        torch.autograd.grad(o, [qkv, table], go)
    tfb = timeit(fb)
    print(f"{name}: windows {B*nW:4d} heads {heads:2d} | fwd(+expand) {tf:7.1f} us | fwd+bwd {tfb:7.1f} us | bwd ~{tfb - tf:7.1f} us")
