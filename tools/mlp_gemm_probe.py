#!/usr/bin/env python3
"""Decomposes the launch time of the stage-2 MLP GEMMs (1800 x 2048 x 512 class: fc1 forward, fc2 data gradient) by epilogue feature, per tile
size (LAVT_GEMM_TILE), in hipGraph replay (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K, ops
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
M, N, Kd = int(os.environ.get("M", 1800)), int(os.environ.get("N", 2048)), int(os.environ.get("KD", 512))
A = torch.randn(M, Kd, device=dev).to(bf); W = (torch.randn(N, Kd, device=dev) * 0.05).to(bf); Wk = W.t().contiguous()
bias = torch.randn(N, device=dev); Cc = torch.empty(M, N, device=dev, dtype=bf); Cp = torch.empty(M, N, device=dev, dtype=bf)
R = torch.randn(M, N, device=dev).to(bf)
wsum = torch.randn(N, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
cases = {
    "plain": dict(),
    "bias": dict(bias=bias),
    "bias+gelu": dict(bias=bias, act=K.ACT_GELU),
    "bias+gelu+Cpre": dict(bias=bias, act=K.ACT_GELU, Cpre=Cp, ldcpre=N),
    "bias+gelu+Cpre+LN": dict(bias=bias, act=K.ACT_GELU, Cpre=Cp, ldcpre=N, ln=(wsum, mean, rstd, 1e-5)),
    "bias+res": dict(bias=bias, R=R, ldr=N),
}
for name, kw in cases.items():
    t = timeit(lambda: ops.gemm_nt(bf, M, N, Kd, A, Kd, W, Kd, Cc, N, **kw))
    print(f"tile {os.environ.get('LAVT_GEMM_TILE', 'auto'):5s} nt {M}x{N}x{Kd} {name:22s} {t * 1e6:7.2f} us  {2.0 * M * N * Kd / t * 1e-12:6.1f} TF/s")
# data gradient with the activation derivative (fc2 dgrad): k-major weight, dact
G = torch.randn(M, Kd, device=dev).to(bf)
for name, kw in {"kmajor": dict(b_kmajor=True), "kmajor+dgelu": dict(b_kmajor=True, dact_pre=Cp, lddact=N, dact=K.ACT_GELU)}.items():
    Wkm = torch.randn(Kd, N, device=dev).to(bf) * 0.05
    t = timeit(lambda: ops.gemm_nt(bf, M, N, Kd, G, Kd, Wkm, N, Cc, N, **kw))
    print(f"tile {os.environ.get('LAVT_GEMM_TILE', 'auto'):5s} nt {M}x{N}x{Kd} {name:22s} {t * 1e6:7.2f} us  {2.0 * M * N * Kd / t * 1e-12:6.1f} TF/s")
