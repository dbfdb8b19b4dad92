#!/usr/bin/env python3
"""How the (window, head) workgroup count of the attention backward meets the 256 CUs: the chained backward launch (no binning launch of its own) for
nwin = 12 .. 32 windows x 16 heads of 144 tokens, hipGraph-timed.  16 windows = 256 workgroups = one per CU; stage 2 at 2 images has 18 = 288.  Run on the GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import _capi as K
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
heads, ws = 16, 12
N, Cc = ws * ws, heads * 32
ld = 160
R = (2 * ws - 1) ** 2
for nwin in (8, 12, 14, 16, 18, 20, 24, 32, 36):
    qkv = torch.randn(nwin * N, 3 * Cc, device=dev).to(bf)
    out = torch.empty(nwin * N, Cc, device=dev, dtype=bf); lse = torch.empty(nwin, heads, N, device=dev)
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv)
    table = torch.randn(R, heads, device=dev) * 0.1
    wsb = torch.empty(int(K.lib.lavt_window_attn_bwd_ws(K.dt(bf), nwin, N, heads, ld, 1, ws, ws)), device=dev)
    pieces = int(K.lib.lavt_window_attn_bwd_pieces(K.dt(bf), nwin, N, heads, ld))
    parts = torch.empty(pieces * heads * R, device=dev)
    K.check(K.lib.lavt_window_attn_fwd(K.dt(bf), K.ptr(qkv), None, ld, None, 0, K.ptr(out), K.ptr(lse), K.ptr(table), 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    mine = K.DtableJob()
    bwd = lambda: K.check(K.lib.lavt_window_attn_bwd_chained(K.dt(bf), K.ptr(qkv), ld, None, 0, K.ptr(out), K.ptr(dout), K.ptr(lse), K.ptr(dqkv), K.ptr(table), K.ptr(wsb), wsb.numel(),
                                                             K.ptr(parts), 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, None, C.byref(mine), K.stream()))
    fwd = lambda: K.check(K.lib.lavt_window_attn_fwd(K.dt(bf), K.ptr(qkv), None, ld, None, 0, K.ptr(out), K.ptr(lse), K.ptr(table), 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    tb, tf = timeit(bwd, iters=10), timeit(fwd, iters=10)
    print(f"nwin {nwin:3d} ({nwin * heads:4d} workgroups = {nwin * heads / 256:.3f} per CU): backward {tb * 1e6:6.1f} us  forward core {tf * 1e6:6.1f} us", flush=True)
