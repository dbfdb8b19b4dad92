#!/usr/bin/env python3
"""fp8 vs bf16 on the decoder convolution shapes (GPU box): the e4m3 implicit GEMM alone, the activation quantiser alone, the bf16 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
for (B, H, Cin, Cout) in ((2, 120, 512, 512), (4, 120, 512, 512), (4, 60, 512, 512), (2, 120, 640, 512)):
    M = B * H * H
    x = torch.randn(M, Cin, device=dev).to(bf)
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5)
    y = torch.empty(M, Cout, device=dev, dtype=bf)
    Wp = ops.weights.get(w, bf, "conv3")
    Wq, wa = ops.weights.get_fp8(w, "conv3")
    xq, ap = ops.fp8.quantize(x, "bench")
    fl = 2.0 * M * Cout * 9 * Cin
    t_bf = timeit(lambda: ops.gemm_nt(bf, M, Cout, 9 * Cin, x, Cin, Wp, 9 * Cin, y, Cout, conv=(H, H, Cin, 0, 1, 1, 3, 3)), iters=10)
    t_f8 = timeit(lambda: ops.gemm_nt(torch.uint8, M, Cout, 9 * Cin, xq, Cin, Wq, 9 * Cin, y, Cout, conv=(H, H, Cin, 0, 1, 1, 3, 3), deq=(ap, wa.data_ptr())), iters=10)
    t_q = timeit(lambda: ops.fp8.quantize(x, "bench"), iters=10)
    print(f"conv {Cin}->{Cout} @{H}x{H} b{B}: bf16 {t_bf*1e6:7.1f} us ({fl/t_bf/1e12:5.0f} TF/s) | fp8 gemm {t_f8*1e6:7.1f} us ({fl/t_f8/1e12:5.0f} TF/s) | quantise {t_q*1e6:6.1f} us")
