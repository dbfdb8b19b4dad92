#!/usr/bin/env python3
"""Times the fused window-attention kernels (forward / backward) through the C ABI at the Swin-B w12 stage shapes (batch 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K
dev, bf = "cuda:0", torch.bfloat16
def run(nwin, heads, ws, iters=20):
    N, C = ws * ws, heads * 32
    ld = 64 if N <= 64 else -(-N // 32) * 32
    qkv = torch.randn(nwin * N, 3 * C, device=dev).to(bf)
    dense = torch.randn(heads, N, ld, device=dev) * 0.1
    dense[:, :, N:] = -1e30
    out = torch.empty(nwin * N, C, device=dev, dtype=bf); lse = torch.empty(nwin, heads, N, device=dev)
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv)
    dtable = torch.zeros((2 * ws - 1) ** 2, heads, device=dev); wsb = torch.empty(int(K.lib.lavt_window_attn_bwd_ws(K.dt(bf), nwin, N, heads, ld, 1, ws, ws)), device=dev); table = torch.randn((2 * ws - 1) ** 2, heads, device=dev) * 0.1
    st = K.stream()
    fwd = lambda: K.check(K.lib.lavt_window_attn_fwd(K.dt(bf), K.ptr(qkv), K.ptr(dense), ld, None, 0, K.ptr(out), K.ptr(lse), K.ptr(table), 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    bwd = lambda: K.check(K.lib.lavt_window_attn_bwd(K.dt(bf), K.ptr(qkv), K.ptr(dense), ld, None, 0, K.ptr(out), K.ptr(dout), K.ptr(lse), K.ptr(dqkv), K.ptr(table), K.ptr(dtable), K.ptr(wsb), wsb.numel(), None, 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    res = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                      # graph-timed: no Python / ctypes time between the launches
        with torch.cuda.graph(g):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / (5 * iters) * 1e3)
    fl = 4.0 * nwin * heads * N * N * 32
    print(f"nwin={nwin:4d} heads={heads:2d} ws={ws:2d}: fwd {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF/s)   bwd {res[1]:7.1f} us ({2.5 * fl / res[1] / 1e6:6.1f} TF/s)")
if len(sys.argv) > 1 and sys.argv[1] == "scan":          # time against the number of (window, head) pairs: one workgroup per CU vs several
    for nwin in (4, 8, 16, 18, 24, 32, 48, 64):
        run(nwin, 16, 12)
else:
    for nwin, heads in ((200, 4), (50, 8), (18, 16), (8, 32)):
        run(nwin, heads, 12)
    run(648, 3, 7); run(72, 12, 7)
