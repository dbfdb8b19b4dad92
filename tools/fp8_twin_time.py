#!/usr/bin/env python3
"""fp8 twins (configs[4]): the producing kernels with and without the e4m3 twin, against the separate quantiser launch they replace (GPU box, hipGraph-timed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import lavt_hip
from lavt_hip import ops, _capi as K
from gemm_bench import timeit
dev, bf = torch.device("cuda:0"), torch.bfloat16
C = 512
for (B, Hi, Ho) in ((4, 30, 60), (4, 60, 120), (2, 60, 120)):
    x = torch.randn(B * Hi * Hi, C, device=dev).to(bf)
    y = torch.empty(B * Ho * Ho, C, device=dev, dtype=bf)
    q = torch.empty(B * Ho * Ho, C, device=dev, dtype=torch.uint8)
    am = torch.zeros(2, device=dev)
    t0 = timeit(lambda: K.check(K.lib.lavt_bilinear_fwd(K.BF16, K.ptr(x), K.ptr(y), B, Hi, Hi, Ho, Ho, C, K.stream())), iters=10)
    t1 = timeit(lambda: K.check(K.lib.lavt_bilinear_fwd_q8(K.ptr(x), K.ptr(y), K.ptr(q), am.data_ptr(), am.data_ptr() + 4, B, Hi, Hi, Ho, Ho, C, K.stream())), iters=10)
    t2 = timeit(lambda: K.check(K.lib.lavt_fp8_quantize(K.BF16, K.ptr(y), K.ptr(q), y.numel(), am.data_ptr(), am.data_ptr() + 4, K.stream())), iters=10)
    print(f"bilinear {B}x{Hi}->{Ho} C={C}: plain {t0*1e6:6.1f} us  with twin {t1*1e6:6.1f} us  separate quantiser {t2*1e6:6.1f} us", flush=True)
for (B, H) in ((4, 60), (4, 120), (2, 120)):
    R = B * H * H
    x = torch.randn(R, C, device=dev).to(bf)
    y = torch.empty_like(x); dy = torch.randn(R, C, device=dev).to(bf); dx = torch.empty_like(x)
    q = torch.empty(R, C, device=dev, dtype=torch.uint8)
    mean, rstd, g, b = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    s = torch.zeros(2, C, device=dev)
    am = torch.zeros(2, device=dev)
    t0 = timeit(lambda: K.check(K.lib.lavt_norm_apply(K.BF16, K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(g), K.ptr(b), None, 1, K.ptr(y), 1, R, C, K.stream())), iters=10)
    t1 = timeit(lambda: K.check(K.lib.lavt_norm_apply_q8(K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(g), K.ptr(b), None, 1, K.ptr(y), K.ptr(q), am.data_ptr(), am.data_ptr() + 4, 1, R, C, K.stream())), iters=10)
    t2 = timeit(lambda: K.check(K.lib.lavt_norm_bwd_apply(K.BF16, K.ptr(dy), K.ptr(x), K.ptr(y), K.ptr(mean), K.ptr(rstd), K.ptr(g), K.ptr(b), None, 1, K.ptr(s[0]), K.ptr(s[1]), float(R), K.ptr(dx), None, 1, R, C, K.stream())), iters=10)
    t3 = timeit(lambda: K.check(K.lib.lavt_norm_bwd_apply_amax(K.ptr(dy), K.ptr(x), K.ptr(y), K.ptr(mean), K.ptr(rstd), K.ptr(g), K.ptr(b), None, 1, K.ptr(s[0]), K.ptr(s[1]), float(R), K.ptr(dx), None, am.data_ptr() + 4, 1, R, C, K.stream())), iters=10)
    print(f"BatchNorm+ReLU {R}x{C}: apply {t0*1e6:6.1f} us  with twin {t1*1e6:6.1f} us | backward apply {t2*1e6:6.1f} us  with |max| {t3*1e6:6.1f} us", flush=True)
