#!/usr/bin/env python3
"""Why do the stage-2 GEMMs take ~2x longer inside the step than alone?  Same kernel, fc1 shape (1800 x 2048 x 512, bf16), under:
  a) the same operands every launch, b) 24 weight buffers in rotation (L2-cold, Infinity-Cache-warm), c) as b with the caches flushed by a
  600 MB fill between launches, d) as a with the step's epilogue (bias + GELU + pre-activation store), e) d + rotation.  GPU box only."""
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
    return e0.elapsed_time(e1) / reps * 1e3          # us per replay

for (M, N, Kd, tag) in ((1800, 2048, 512, "fc1"), (2592, 1536, 512, "qkv"), (1800, 512, 2048, "fc2"), (2592, 512, 512, "proj")):
    NW = 24
    A = [torch.randn(M, Kd, device=dev).to(bf) for _ in range(NW)]
    W = [torch.randn(N, Kd, device=dev).to(bf) for _ in range(NW)]
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf); P = torch.empty_like(C)
    flush = torch.empty(300 * 1024 * 1024 // 4, device=dev)
    plain = lambda i: (lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[i], Kd, C, N))
    rotA = lambda i: (lambda: ops.gemm_nt(bf, M, N, Kd, A[i], Kd, W[i], Kd, C, N))
    epi = lambda i: (lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[i], Kd, C, N, bias=bias, act=K.ACT_GELU, Cpre=P, ldcpre=N))
    R = torch.randn(M, N, device=dev).to(bf)
    rmap = torch.randperm(M, device=dev).to(torch.int32)
    rs = torch.ones(2, device=dev)
    eb = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, bias=bias)
    er = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, bias=bias, R=R, ldr=N)
    es = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, bias=bias, R=R, ldr=N, c_rowmap=rmap, row_scale=rs, row_scale_div=(M + 1) // 2)
    eg = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, a_rowmap=rmap, bias=bias)
    fl = lambda: flush.fill_(1.0)
    n = 48
    ta = graph_time([plain(0)] * n) / n
    tb = graph_time([plain(i % NW) for i in range(n)]) / n
    tb2 = graph_time([rotA(i % NW) for i in range(n)]) / n
    tf = graph_time([fl] * 12) / 12
    tc = graph_time([f for i in range(12) for f in (fl, rotA(i % NW))]) / 12 - tf
    td = graph_time([epi(0)] * n) / n
    te = graph_time([epi(i % NW) for i in range(n)]) / n
    ea = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, bias=bias, act=K.ACT_GELU)                      # GELU, one output
    ep = lambda: ops.gemm_nt(bf, M, N, Kd, A[0], Kd, W[0], Kd, C, N, bias=bias, Cpre=P, ldcpre=N)                    # two outputs, no activation
    t2 = [graph_time([f] * n) / n for f in (ea, ep)]
    print(f"{tag:5s}   bias+GELU (one output) {t2[0]:5.1f} | bias + second output (no activation) {t2[1]:5.1f}")
    tv = [graph_time([f] * n) / n for f in (eb, er, es, eg)]
    print(f"{tag:5s}   epilogues: bias {tv[0]:5.1f} | bias+residual {tv[1]:5.1f} | bias+residual+scatter+rowscale {tv[2]:5.1f} | gathered A + bias {tv[3]:5.1f}")
    print(f"{tag:5s} {M}x{N}x{Kd}: same operands {ta:5.1f} us | 24 weights in rotation {tb:5.1f} | weights+inputs in rotation {tb2:5.1f} | after a 1.2 GB cache flush {tc:5.1f} (flush {tf:.0f}) | bias+GELU+pre epilogue {td:5.1f} | epilogue + rotation {te:5.1f}")
