#!/usr/bin/env python3
"""Stage-1/2/3 grouped weight-gradient launches in token order (five members): the 64x64-tile grouped launch against the stream-K form
(lavt_gemm_tn_grouped_sk), hipGraph-timed.  LAVT_PROBE=<runs> sets the number of persistent workgroups.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import _capi as K, ops, rowmaps
from gemm_bench import timeit
dev, bf = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(3)
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)
for (B, H, ws, shift, Cc) in ((2, 30, 12, 6, 512), (4, 30, 12, 6, 512), (2, 60, 12, 6, 256), (2, 15, 12, 6, 1024)):
    inv, pad = rowmaps.window_inverse(B, H, H, ws, shift, dev), rowmaps.window_pad_rows(B, H, H, ws, shift, dev)
    T, Mw = B * H * H, B * (-(-H // ws) * ws) ** 2
    dqkv, xn, dy, o = mk(Mw, 3 * Cc), mk(T, Cc), mk(T, Cc), mk(Mw, Cc)
    dpre, x2, dy2, h = mk(T, 4 * Cc), mk(T, Cc), mk(T, Cc), mk(T, 4 * Cc)
    structs, keep = [], []
    class Q:
        def add(self, p, t, extra=False): structs.append(p); keep.append(t)
    outs = [torch.zeros(Cc, 4 * Cc, device=dev), torch.zeros(4 * Cc, Cc, device=dev), torch.zeros(Cc, Cc, device=dev), torch.zeros(3 * Cc, Cc, device=dev)]
    bs = [torch.zeros(Cc, device=dev), torch.zeros(4 * Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(3 * Cc, device=dev)]
    ops.gemm_tn(bf, Cc, 4 * Cc, T, dy2, Cc, h, 4 * Cc, outs[0], 4 * Cc, colsum=bs[0], defer=Q())
    ops.gemm_tn(bf, 4 * Cc, Cc, T, dpre, 4 * Cc, x2, Cc, outs[1], Cc, colsum=bs[1], defer=Q())
    ops.gemm_tn(bf, Cc, Cc, T, dy, Cc, o, Cc, outs[2], Cc, b_rowmap=inv, colsum=bs[2], defer=Q())
    dummy = torch.empty(3 * Cc, 8, device=dev)
    if pad.numel():
        ops.gemm_tn(bf, 3 * Cc, 8, pad.numel(), dqkv, 3 * Cc, ops._zero_page_tensor(dev), 0, dummy, 8, a_rowmap=pad, colsum=bs[3], colsum_atomic=True, defer=Q(), extra=True)
    ops.gemm_tn(bf, 3 * Cc, Cc, T, dqkv, 3 * Cc, xn, Cc, outs[3], Cc, a_rowmap=inv, colsum=bs[3], colsum_atomic=True, defer=Q())
    ops.assign_partials(structs, dev)
    arr = (K.GemmTN * len(structs))(*structs)
    n = len(structs)
    flops = sum(2.0 * q.I * q.J * q.K for q in structs)
    need = int(K.lib.lavt_gemm_tn_grouped_sk_ws(arr, n))
    old = lambda: K.check(K.lib.lavt_gemm_tn_grouped(arr, n, K.stream()))
    for t in outs + bs: t.zero_()
    old(); torch.cuda.synchronize(); ref = [t.clone() for t in outs]
    line = f"group B{B} {H}x{H} C{Cc} ({flops/1e9:.2f} GFLOP): grouped 64x64 {timeit(old, iters=10)*1e6:6.1f} us"
    if need:
        scr = torch.empty(need, device=dev)
        new = lambda: K.check(K.lib.lavt_gemm_tn_grouped_sk(arr, n, K.ptr(scr), scr.numel(), K.stream()))
        for t in outs + bs: t.zero_()
        new(); torch.cuda.synchronize()
        err = max(float((a - r).abs().max() / r.abs().max()) for a, r in zip(outs, ref))
        tn = timeit(new, iters=10)
        line += f" | stream-K 128x128 {tn*1e6:6.1f} us ({flops/tn/2.5e15:.3f} of peak) rel diff {err:.1e} scratch {need*4/1e6:.0f} MB"
    else:
        line += " | stream-K: not applicable"
    print(line, flush=True)
