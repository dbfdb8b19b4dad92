#!/usr/bin/env python3
"""Runs the one-kernel W-MSA forward (norm1 + window gather + qkv + attention, csrc/wmsa_fused.hip) of a stage-2 Swin-B block at batch 2
(shifted windows) a few times: target of the rocprofv3 --pmc passes (tools/pmc_passes.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops, rowmaps
from lib.backbone import SwinTransformerBlock
dev = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16)
C, H, ws, B = 512, int(os.environ.get("WMSA_H", 30)), 12, int(os.environ.get("WMSA_B", 2))
blk = SwinTransformerBlock(C, C // 32, ws, shift_size=ws // 2).to(dev)
x = torch.randn(B * H * H, C, device=dev).to(torch.bfloat16)
wmap = rowmaps.window_map(B, H, H, ws, blk.shift_size, dev)
region = rowmaps.region_ids(H, H, ws, blk.shift_size, dev)
with torch.no_grad():
    for _ in range(6):
        ops.wmsa_fused(x, blk.norm1, blk.attn, region, wmap, ws, C // 32)
torch.cuda.synchronize()
Mw = wmap.numel()
print("flops_per_launch", 2.0 * Mw * 3 * C * C + 4.0 * (Mw // 144) * (C // 32) * 144 * 144 * 32)
