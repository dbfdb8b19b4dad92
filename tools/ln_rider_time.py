#!/usr/bin/env python3
"""Stage-2 grouped weight-gradient launch + norm1's LayerNorm backward: as two launches against lavt_gemm_tn_grouped_ln (the LayerNorm as rider
workgroups), hipGraph-timed.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import _capi as K, ops, rowmaps
from gemm_bench import timeit
dev, bf = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(3)
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)
CFGS = ((2, 30, 12, 6, 512), (2, 60, 12, 6, 256), (2, 120, 12, 6, 128))
if os.environ.get("TNP_SCALING"):          # the stage-2 group at 2 / 4 / 6 / 8 images: time against K = 1800 .. 7200 rows (slope = us per K tile, intercept = the launch's fixed cost)
    CFGS = tuple((b, 30, 12, 6, 512) for b in (2, 4, 6, 8))
for (B, H, ws, shift, Cc) in CFGS:
    inv, pad = rowmaps.window_inverse(B, H, H, ws, shift, dev), rowmaps.window_pad_rows(B, H, H, ws, shift, dev)
    T, Mw = B * H * H, B * (-(-H // ws) * ws) ** 2
    dqkv, xn, dy, o = mk(Mw, 3 * Cc), mk(T, Cc), mk(T, Cc), mk(Mw, Cc)
    dpre, x2, dy2, h = mk(T, 4 * Cc), mk(T, Cc), mk(T, Cc), mk(T, 4 * Cc)
    structs, keep = [], []
    class Q:
        def add(self, p, t, extra=False, rider=None): structs.append(p); keep.append(t)
    outs = [torch.zeros(Cc, 4 * Cc, device=dev), torch.zeros(4 * Cc, Cc, device=dev), torch.zeros(Cc, Cc, device=dev), torch.zeros(3 * Cc, Cc, device=dev)]
    bs = [torch.zeros(Cc, device=dev), torch.zeros(4 * Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(3 * Cc, device=dev)]
    ops.gemm_tn(bf, Cc, 4 * Cc, T, dy2, Cc, h, 4 * Cc, outs[0], 4 * Cc, colsum=bs[0], defer=Q())
    ops.gemm_tn(bf, 4 * Cc, Cc, T, dpre, 4 * Cc, x2, Cc, outs[1], Cc, colsum=bs[1], defer=Q())
    ops.gemm_tn(bf, Cc, Cc, T, dy, Cc, o, Cc, outs[2], Cc, b_rowmap=inv, colsum=bs[2], defer=Q())
    if pad.numel():
        dummy = torch.empty(3 * Cc, 8, device=dev)
        ops.gemm_tn(bf, 3 * Cc, 8, pad.numel(), dqkv, 3 * Cc, ops._zero_page_tensor(dev), 0, dummy, 8, a_rowmap=pad, colsum=bs[3], colsum_atomic=True, defer=Q(), extra=True)
    ops.gemm_tn(bf, 3 * Cc, Cc, T, dqkv, 3 * Cc, xn, Cc, outs[3], Cc, a_rowmap=inv, colsum=bs[3], colsum_atomic=True, defer=Q())
    ops.assign_partials(structs, dev)
    arr = (K.GemmTN * len(structs))(*structs)
    n = len(structs)
    dxn, x, dres = mk(T, Cc), mk(T, Cc), mk(T, Cc)
    gamma = torch.ones(Cc, device=dev)
    mean, rstd = x.float().mean(1), torch.rsqrt(x.float().var(1, unbiased=False) + 1e-5)
    nblk = int(K.lib.lavt_layernorm_bwd_blocks(K.BF16, T, Cc))
    dx = torch.empty_like(x); wsl = torch.empty(nblk * 2 * Cc, device=dev)
    grp = lambda: K.check(K.lib.lavt_gemm_tn_grouped(arr, n, K.stream()))
    ln = lambda: K.check(K.lib.lavt_layernorm_bwd_partial(K.BF16, K.ptr(dxn), K.ptr(x), None, K.ptr(gamma), K.ptr(mean), K.ptr(rstd), K.ptr(dx), K.ptr(wsl), wsl.numel(), K.ptr(dres), T, Cc, K.stream()))
    def two(): grp(); ln()
    one = lambda: K.check(K.lib.lavt_gemm_tn_grouped_ln(arr, n, K.ptr(dxn), K.ptr(x), K.ptr(gamma), K.ptr(mean), K.ptr(rstd), K.ptr(dx), K.ptr(wsl), wsl.numel(), K.ptr(dres), T, Cc, K.stream()))
    print(f"B{B} {H}x{H} C{Cc}: group {timeit(grp, iters=10)*1e6:6.1f} us  LN bwd {timeit(ln, iters=10)*1e6:5.1f} us  two launches {timeit(two, iters=10)*1e6:6.1f} us  |  with LN riders {timeit(one, iters=10)*1e6:6.1f} us  (LN blocks {nblk})", flush=True)
