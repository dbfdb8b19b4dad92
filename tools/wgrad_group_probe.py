#!/usr/bin/env python3
"""What the stage-2 grouped weight-gradient launch spends its time on: the same four problems with / without row maps, column sums, split chains."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import _capi as K, ops
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
g = torch.Generator().manual_seed(21)
M, Mw, Cc = 1800, 2592, 512
wmap = torch.randint(0, M, (Mw,), generator=g, dtype=torch.int32).to(dev)
mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(dev).to(bf)

def build(maps=True, colsum=True, win_rows=Mw):
    probs = [(4 * Cc, Cc, M, mk(M, 4 * Cc), mk(M, Cc), {}), (Cc, 4 * Cc, M, mk(M, Cc), mk(M, 4 * Cc), {}),
             (3 * Cc, Cc, win_rows, mk(win_rows, 3 * Cc), mk(M if maps else win_rows, Cc), dict(b_rowmap=wmap[:win_rows]) if maps else {}),
             (Cc, Cc, win_rows, mk(M if maps else win_rows, Cc), mk(win_rows, Cc), dict(a_rowmap=wmap[:win_rows]) if maps else {})]
    structs, keep = [], []
    class Q:
        def add(self, p, t, extra=False): structs.append(p); keep.append(t)
    for I, J, Kd, A, B, kw in probs:
        out = torch.zeros(I, J, device=dev); cs = torch.zeros(I, device=dev) if colsum else None
        ops.gemm_tn(bf, I, J, Kd, A, I, B, J, out, J, colsum=cs, defer=Q(), **kw)
        keep.append((out, cs))
    arr = (K.GemmTN * len(structs))(*structs)
    return lambda: K.check(K.lib.lavt_gemm_tn_grouped(arr, len(structs), K.stream())), keep

for name, kw in (("as in the step (maps on qkv/proj, colsum)", {}), ("no row maps", dict(maps=False)), ("no colsum", dict(colsum=False)),
                 ("no maps, no colsum", dict(maps=False, colsum=False)), ("window rows 1800 (no padded rows), maps", dict(win_rows=1800)),
                 ("window rows 1800, no maps, no colsum", dict(win_rows=1800, maps=False, colsum=False))):
    fn, keep = build(**kw)
    print(f"{name:48s} {timeit(fn) * 1e6:6.1f} us")
