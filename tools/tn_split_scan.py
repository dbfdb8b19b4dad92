#!/usr/bin/env python3
"""Long-K weight gradients on few output tiles (PWAM 1x1 convolutions): time against the number of K pieces, with partial tiles and with atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops
dev, bf = "cuda:0", torch.bfloat16
def gt(fn, n=20, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
for I, J, Kd in ((128, 128, 28800), (256, 256, 7200), (512, 128, 28800), (128, 768, 28800), (512, 512, 1800), (1024, 256, 7200)):
    A = torch.randn(Kd, I, device=dev).to(bf); B = torch.randn(Kd, J, device=dev).to(bf)
    out = torch.zeros(I, J, device=dev); cs = torch.zeros(I, device=dev)
    res = []
    for parts in ("1", "0"):
        os.environ["LAVT_TN_PARTIALS"] = parts
        for sp in (0, 4, 8, 16, 32, 57, 113):
            if sp: os.environ["LAVT_TN_SPLIT"] = str(sp)
            else: os.environ.pop("LAVT_TN_SPLIT", None)
            t = gt(lambda: ops.gemm_tn(bf, I, J, Kd, A, I, B, J, out, J, colsum=cs, accumulate=True))
            res.append(f"{'P' if parts == '1' else 'A'}s{sp or 'auto'}: {t:5.1f}")
    os.environ.pop("LAVT_TN_SPLIT", None)
    print(f"{I}x{J}x{Kd}: " + " | ".join(res))
