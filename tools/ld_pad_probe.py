#!/usr/bin/env python3
"""Does a power-of-two row pitch of the operands cost bandwidth?  Times the stage-2 grouped weight-gradient problem (and one NT GEMM) with
contiguous operands (pitch = columns) against operands whose rows are padded by 64 / 128 bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import ctypes as C
import torch
from lavt_hip import ops
from lavt_hip import _capi as K
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
Krows = 2592
for pad in (0, 32, 64, 8):
    probs, keep = [], []
    for (I, J) in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        A = torch.randn(Krows, I + pad, device=dev).to(bf); B = torch.randn(Krows, J + pad, device=dev).to(bf)
        Cc = torch.zeros(I, J, device=dev)
        class Q:  # capture the struct the wrapper builds
            items = []
            def add(self, p, t): self.items.append(p); keep.append(t)
        q = Q(); q.items = []
        ops.gemm_tn(bf, I, J, Krows, A, I + pad, B, J + pad, Cc, J, defer=q)
        probs.append(q.items[0])
    arr = (K.GemmTN * 4)(*probs)
    fn = lambda: K.check(K.lib.lavt_gemm_tn_grouped(arr, 4, K.stream()))
    t = timeit(fn, iters=10)
    fl = 2.0 * Krows * sum(i * j for i, j in ((1536, 512), (512, 512), (2048, 512), (512, 2048)))
    print(f"grouped wgrad stage 2, row pad {pad:3d} elements: {t*1e6:7.1f} us  {fl/t/1e12:6.0f} TF/s")
for pad in (0, 32, 64):
    M, N, Kd = 2592, 512, 512
    A = torch.randn(M, Kd + pad, device=dev).to(bf); B = torch.randn(N, Kd + pad, device=dev).to(bf); Cc = torch.empty(M, N, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_nt(bf, M, N, Kd, A, Kd + pad, B, Kd + pad, Cc, N), iters=20)
    print(f"NT 2592x512x512, row pad {pad:3d}: {t*1e6:7.1f} us")
for pad in (0, 32, 64):
    M, N, Kd = 2592, 2048, 512
    A = torch.randn(M, Kd + pad, device=dev).to(bf); B = torch.randn(N, Kd + pad, device=dev).to(bf); Cc = torch.empty(M, N, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_nt(bf, M, N, Kd, A, Kd + pad, B, Kd + pad, Cc, N), iters=20)
    print(f"NT 2592x2048x512, row pad {pad:3d}: {t*1e6:7.1f} us")
