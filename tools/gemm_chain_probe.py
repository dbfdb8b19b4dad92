#!/usr/bin/env python3
"""Does a thin GEMM pay for reading what the previous kernel has just written?  PWAM shapes (M x C x C, bf16): a chain Y1 = X W, Y2 = Y1 W, ...
(every launch reads the previous launch's output) against the same launches all reading one static X.  hipGraph-timed.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops, _capi as K
dev, bf = "cuda:0", torch.bfloat16

def graph_time(fns, reps=5):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for M, C in ((28800, 128), (7200, 256), (1800, 512), (1800, 2048)):
    n = 24
    X = torch.randn(M, C, device=dev).to(bf) * 0.05
    W = (torch.randn(C, C, device=dev) * C ** -0.5).to(bf)
    Y = [torch.empty(M, C, device=dev, dtype=bf) for _ in range(n + 1)]
    bias = torch.zeros(C, device=dev)
    static = [(lambda i=i: ops.gemm_nt(bf, M, C, C, X, C, W, C, Y[i + 1], C, bias=bias)) for i in range(n)]
    def link(i):
        src = X if i == 0 else Y[i]
        return lambda: ops.gemm_nt(bf, M, C, C, src, C, W, C, Y[i + 1], C, bias=bias)
    chain = [link(i) for i in range(n)]
    # the same chain through 2 ping-pong buffers only (output lines are the ones read one launch ago)
    P = [torch.empty(M, C, device=dev, dtype=bf) for _ in range(2)]
    P[0].copy_(X)
    pp = [(lambda i=i: ops.gemm_nt(bf, M, C, C, P[i & 1], C, W, C, P[(i + 1) & 1], C, bias=bias)) for i in range(n)]
    ts, tc, tp = graph_time(static) / n, graph_time(chain) / n, graph_time(pp) / n
    print(f"M={M:6d} C={C:5d}: static input {ts:5.1f} us | chained through {n} buffers {tc:5.1f} us | ping-pong {tp:5.1f} us   ({2 * M * C * 2 / 1e6:.1f} MB per launch)")
