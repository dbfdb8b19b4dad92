#!/usr/bin/env python3
"""Times the conv weight-gradient GEMM (TN family with 3x3 tap gather) at the decoder shapes, per tile / split configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from lavt_hip import ops
from gemm_bench import timeit
dev, bf = "cuda:0", torch.bfloat16
for (B, H, W, Cin, Cout) in ((2, 120, 120, 512, 512), (2, 120, 120, 640, 512), (2, 60, 60, 768, 512)):
    M = B * H * W
    dy = torch.randn(M, Cout, device=dev).to(bf); x = torch.randn(M, Cin, device=dev).to(bf)
    dW = torch.zeros(Cout, Cin * 9, device=dev)
    fn = lambda: ops.gemm_tn(bf, Cout, 9 * Cin, M, dy, Cout, x, Cin, dW, 9 * Cin, conv=(H, W, Cin, 1, 1, 3, 3), c_conv_permute=(os.environ.get("PERMUTE", "0") == "1"))
    res = []
    ref = None
    for tile in ("64", "128"):
        for sp in ("0", "2", "4", "7"):
            os.environ["LAVT_GEMM_TILE"] = tile
            if sp != "0": os.environ["LAVT_TN_SPLIT"] = sp
            else: os.environ.pop("LAVT_TN_SPLIT", None)
            dW.zero_(); fn(); torch.cuda.synchronize()
            if ref is None: ref = dW.clone()
            err = float((dW - ref).abs().max() / ref.abs().max())
            assert err < 2e-3, (tile, sp, err)
            t = timeit(fn, iters=5)
            res.append(f"t{tile}/s{sp if sp != '0' else 'auto'}: {t*1e6:6.1f}us {2.0*M*Cout*9*Cin/t/1e12:4.0f}TF")
    os.environ.pop("LAVT_GEMM_TILE"); os.environ.pop("LAVT_TN_SPLIT", None)
    print(f"conv wgrad {B}x{H}x{W} {Cin}->{Cout} | " + " | ".join(res))
