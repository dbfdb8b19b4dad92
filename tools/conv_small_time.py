#!/usr/bin/env python3
"""Times the decoder's small-pixel-count 3x3 convolutions (forward and data gradient) with / without the tap-split reduction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops
dev = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16)
def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)
def timeit_eager(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (B, H, C1, C2, Cout) in ((2, 30, 1024, 512, 512), (2, 30, 512, 0, 512), (2, 60, 512, 256, 512), (2, 60, 512, 0, 512)):
    M = B * H * H
    x1 = torch.randn(M, C1, device=dev).to(torch.bfloat16)
    x2 = torch.randn(M, C2, device=dev).to(torch.bfloat16) if C2 else None
    w = torch.nn.Parameter(torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02)
    dy = torch.randn(M, Cout, device=dev).to(torch.bfloat16)
    res = {}
    for rows in (0, 1 << 20):
        ops._CONV_SPLIT_MAX_ROWS = rows
        x1r = x1.clone().requires_grad_(True)
        x2r = x2.clone().requires_grad_(True) if C2 else None
        y = ops.conv3x3(x1r, x2r, w, B, H, H)
        def fwd():
            with torch.no_grad():
                ops.conv3x3(x1, x2, w, B, H, H)
        def fb():
            yy = ops.conv3x3(x1r, x2r, w, B, H, H)
            yy.backward(dy)
        tf, tfb = timeit(fwd), timeit_eager(fb)
        res[rows] = (tf, tfb, y.detach().float(), x1r.grad.float().clone())
        x1r.grad = None
    e_y = float((res[0][2] - res[1 << 20][2]).abs().max() / res[0][2].abs().max())
    e_dx = float((res[0][3] - res[1 << 20][3]).abs().max() / res[0][3].abs().max())
    print(f"M={M:5d} Cin={C1}+{C2} Cout={Cout}: fwd {res[0][0]:6.1f} -> {res[1 << 20][0]:6.1f} us   fwd+bwd {res[0][1]:6.1f} -> {res[1 << 20][1]:6.1f} us   rel diff y {e_y:.1e} dx {e_dx:.1e}")
