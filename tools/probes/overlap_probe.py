#!/usr/bin/env python3
"""Feasibility probe (round 6): can the grouped weight-gradient launch of a stage-2 Swin-B block (32.8 us of the block's 110 us backward) run UNDER the
data-gradient chain of the next block when it sits on a forked branch of the captured graph?  Serial graph (chain, wgrad, chain, wgrad, ...) against
a forked one (wgrad of block i on a side stream beside the chain of block i+1; joined at the end).  Prints us per block for both."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops, rowmaps
dev, bf = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(21)
B, H, ws, shift, Cc = 2, 30, 12, 6, 512
inv, pad = rowmaps.window_inverse(B, H, H, ws, shift, dev), rowmaps.window_pad_rows(B, H, H, ws, shift, dev)
wmap = rowmaps.window_map(B, H, H, ws, shift, dev)
M, Mw = B * H * H, wmap.numel()
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)
NBLK = 6


def make_wgrad():
    structs, keep = [], []
    class Q:
        def add(self, p, t, extra=False): structs.append(p); keep.append(t)
    def member(I, J, Kd, A, Bm, **kw):
        out = torch.zeros(I, J, device=dev); cs = kw.pop("cs", None)
        cs = torch.zeros(I, device=dev) if cs is None else cs
        ops.gemm_tn(bf, I, J, Kd, A, I, Bm, kw.pop("ldb", J), out, J, colsum=cs, defer=Q(), **kw)
        keep.append((out, cs))
        return cs
    member(Cc, 4 * Cc, M, mk(M, Cc), mk(M, 4 * Cc))
    member(4 * Cc, Cc, M, mk(M, 4 * Cc), mk(M, Cc))
    dqkv = mk(Mw, 3 * Cc)
    member(Cc, Cc, M, mk(M, Cc), mk(Mw, Cc), b_rowmap=inv)
    bq = member(3 * Cc, 8, pad.numel(), dqkv, ops._zero_page_tensor(dev), ldb=0, a_rowmap=pad, colsum_atomic=True, extra=True)
    member(3 * Cc, Cc, M, dqkv, mk(M, Cc), a_rowmap=inv, colsum_atomic=True, cs=bq)
    ops.assign_partials(structs, dev) if False else None
    arr = (K.GemmTN * len(structs))(*structs)
    return arr, len(structs), keep


wg = [make_wgrad() for _ in range(NBLK)]
# the data-gradient chain of a block: fc2 dgrad (2048 -> ... ), fc1 dgrad, proj dgrad, attention backward, qkv dgrad
W2, W1, Wp, Wq = mk(Cc, 4 * Cc), mk(4 * Cc, Cc), mk(Cc, Cc), mk(3 * Cc, Cc)
dy, h1, d1, d2, d3 = mk(M, Cc), mk(M, 4 * Cc), mk(M, 4 * Cc), mk(M, Cc), mk(Mw, Cc)
nwin, heads, N = Mw // 144, 16, 144
ld = 160
qkv = mk(Mw, 3 * Cc); out = mk(Mw, Cc); lse = torch.randn(nwin, heads, N, device=dev); dqkv = torch.empty_like(qkv)
table = torch.randn(23 * 23, heads, device=dev) * 0.1; dtable = torch.zeros(23 * 23, heads, device=dev)
wsb = torch.empty(int(K.lib.lavt_window_attn_bwd_ws(K.dt(bf), nwin, N, heads, ld, 1, ws, ws)), device=dev)
dx = mk(M, Cc)


def chain():
    ops.gemm_nt(bf, M, 4 * Cc, Cc, dy, Cc, W2, 4 * Cc, d1, 4 * Cc, b_kmajor=True, dact_pre=h1, lddact=4 * Cc, dact=K.ACT_GELU)
    ops.gemm_nt(bf, M, Cc, 4 * Cc, d1, 4 * Cc, W1, Cc, d2, Cc, b_kmajor=True)
    ops.gemm_nt(bf, Mw, Cc, Cc, d2, Cc, Wp, Cc, d3, Cc, b_kmajor=True, a_rowmap=wmap)
    K.check(K.lib.lavt_window_attn_bwd(K.dt(bf), K.ptr(qkv), None, ld, None, 0, K.ptr(out), K.ptr(d3), K.ptr(lse), K.ptr(dqkv), K.ptr(table), K.ptr(dtable), K.ptr(wsb), wsb.numel(),
                                       None, 1, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    ops.gemm_nt(bf, M, Cc, 3 * Cc, dqkv, 3 * Cc, Wq, Cc, dx, Cc, b_kmajor=True, a_rowmap=inv)


def wgrad(i):
    arr, n, _ = wg[i]
    K.check(K.lib.lavt_gemm_tn_grouped(arr, n, K.stream()))


side = torch.cuda.Stream()


def serial():
    for i in range(NBLK):
        chain(); wgrad(i)


def forked():
    main = torch.cuda.current_stream()
    for i in range(NBLK):
        chain()
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad(i)
    main.wait_stream(side)


def only_chain():
    for i in range(NBLK):
        chain()


def only_wgrad():
    for i in range(NBLK):
        wgrad(i)


def timeit(fn, name):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
    torch.cuda.synchronize()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / 20 / NBLK * 1e3:8.1f} us per block")


for name, fn in (("chain", only_chain), ("wgrad", only_wgrad), ("serial", serial), ("forked", forked)):
    timeit(fn, name)
