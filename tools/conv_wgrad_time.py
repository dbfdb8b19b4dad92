#!/usr/bin/env python3
"""Decoder conv weight gradients: the fused-tap kernel (csrc/conv_wgrad.hip) against the tap-shifted TN GEMM + zero fill + unpack it replaces,
hipGraph-timed at the decoder shapes of Swin-B w12 480 (batch 2 and 4).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import ops, _capi as K
from gemm_bench import timeit
dev, bf = torch.device("cuda:0"), torch.bfloat16
shapes = [(2, 120, 120, 512, 0, 512), (2, 120, 120, 512, 128, 512), (2, 60, 60, 512, 256, 512), (2, 30, 30, 1024, 512, 512), (4, 120, 120, 512, 0, 512), (4, 120, 120, 512, 128, 512),
          (4, 60, 60, 512, 256, 512), (4, 60, 60, 512, 0, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, H, W, C1, C2, Cout) in shapes:
    M, Cin = B * H * W, C1 + C2
    dy = torch.randn(M, Cout, device=dev).to(bf)
    x1 = torch.randn(M, C1, device=dev).to(bf)
    x2 = torch.randn(M, C2, device=dev).to(bf) if C2 else None
    dW = torch.zeros(Cout, Cin * 9, device=dev)

    def old():
        packed = torch.zeros(Cout, 9 * Cin, dtype=torch.float32, device=dev)
        ops.gemm_tn(bf, Cout, 9 * Cin, M, dy, Cout, x1, C1, packed, 9 * Cin, B2=x2, ldb2=C2, b_split=C1, conv=(H, W, Cin, 1, 1, 3, 3))
        K.check(K.lib.lavt_unpack_conv_grad(K.ptr(packed), K.ptr(dW), Cout, Cin, 9, K.stream()))
    ws = int(K.lib.lavt_conv3x3_wgrad_ws(B, H, W, Cout, Cin, C1 if C2 else Cin))
    scr = ops._tn_parts(ws, dev)

    def new():
        K.check(K.lib.lavt_conv3x3_wgrad(K.ptr(dy), Cout, K.ptr(x1), C1, K.ptr(x2), C2, C1, B, H, W, Cout, Cin, K.ptr(scr), scr.numel(), K.ptr(dW), 0, ops._zero_page(dev), K.stream()))
    f8ok = K.lib.lavt_conv3x3_wgrad_f8_ok(B, H, W, Cout, Cin, C1 if C2 else Cin) == 1
    if f8ok:                                                  # e4m3 operands (the copies the forward convolution / the data gradient hold in fp8 mode)
        dyq, dya = ops.fp8.quantize_current(dy, "wgrad-bench-dy")
        x1q, xa = ops.fp8.quantize_current(x1, "wgrad-bench-x")
        x2q = ops.fp8.quantize_current(x2, "wgrad-bench-x2")[0] if C2 else None

    def new8():
        K.check(K.lib.lavt_conv3x3_wgrad_f8(K.ptr(dyq), Cout, dya, K.ptr(x1q), C1, K.ptr(x2q), C2, xa, C1, B, H, W, Cout, Cin, K.ptr(scr), scr.numel(), K.ptr(dW), 0,
                                            ops._zero_page(dev), K.stream()))
    dW.zero_(); old(); torch.cuda.synchronize(); ref = dW.clone()
    dW.zero_(); new(); torch.cuda.synchronize()
    err = float((dW - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * Cout * 9 * Cin
    to, tn = timeit(old, iters=5), timeit(new, iters=5)
    t8 = timeit(new8, iters=5) if f8ok else float("nan")
    print(f"conv wgrad {B}x{H}x{W} {C1}+{C2}->{Cout}: e4m3 fused taps {t8*1e6:7.1f} us ({fl/t8/5e15:.3f} of the fp8 peak)", flush=True)
    print(f"conv wgrad {B}x{H}x{W} {C1}+{C2}->{Cout}: old {to*1e6:7.1f} us ({fl/to/2.5e15:.3f} of peak)  fused taps {tn*1e6:7.1f} us ({fl/tn/2.5e15:.3f})  rel diff {err:.1e}  scratch {ws*4/1e6:.0f} MB", flush=True)
