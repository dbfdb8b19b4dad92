#!/usr/bin/env python3
"""Times norm1 + qkv + attention of a Swin block (forward) as the unfused kernel sequence and as the one-kernel form, per stage shape of Swin-B w12 480^2 batch 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops, rowmaps
from lib.backbone import SwinTransformerBlock
dev = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16)
for C, H, shifted in ((128, 120, 1), (256, 60, 1), (512, 30, 1), (512, 30, 0), (1024, 15, 1)):
    ws, nH, B = 12, C // 32, 2
    blk = SwinTransformerBlock(C, nH, ws, shift_size=ws // 2 if shifted else 0).to(dev)
    x = torch.randn(B * H * H, C, device=dev).to(torch.bfloat16)
    wmap = rowmaps.window_map(B, H, H, ws, blk.shift_size, dev)
    region = rowmaps.region_ids(H, H, ws, blk.shift_size, dev) if shifted else None
    a = blk.attn
    def unfused():
        xn, _ = ops.layer_norm_res(x, blk.norm1.weight, blk.norm1.bias, 1e-5)
        qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias, in_map=wmap, rows=wmap.numel())
        return ops.window_attention(qkv, a.relative_position_bias_table, region, ws, nH)
    def fused():
        return ops.wmsa_fused(x, blk.norm1, a, region, wmap, ws, nH)[0]
    res = []
    with torch.no_grad():
        o1, o2 = unfused(), fused()
        err = float((o1.float() - o2.float()).abs().max())
        for fn in (unfused, fused):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(20):
                        fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); torch.cuda.synchronize()
            e0.record(); [g.replay() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 100)
    print(f"C={C:5d} tokens={B * H * H:6d} windows x heads={wmap.numel() // 144 * nH:5d}  unfused {res[0]:6.1f} us   fused {res[1]:6.1f} us   max|diff| {err:.3f}")
