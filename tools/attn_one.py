#!/usr/bin/env python3
"""Runs the fused window-attention forward / backward of a stage-2 Swin-B w12 block at batch 2 (18 windows x 16 heads, 144 tokens) a few times:
target of the rocprofv3 --pmc passes (tools/pmc_passes.sh attn_one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import _capi as K
dev, bf = "cuda:0", torch.bfloat16
nwin, heads, ws, wd = int(os.environ.get("ATTN_NWIN", 18)), int(os.environ.get("ATTN_HEADS", 16)), int(os.environ.get("ATTN_WS", 12)), int(os.environ.get("ATTN_WD", 1))
N, C = wd * ws * ws, heads * 32          # ATTN_WD=8 ATTN_WS=7: the 392-token window of Video-Swin
ld = -(-N // 32) * 32
qkv = torch.randn(nwin * N, 3 * C, device=dev).to(bf)
out = torch.empty(nwin * N, C, device=dev, dtype=bf); lse = torch.empty(nwin, heads, N, device=dev)
dout = torch.randn_like(out); dqkv = torch.empty_like(qkv)
R = (2 * wd - 1) * (2 * ws - 1) ** 2
dtable = torch.zeros(R, heads, device=dev)
wsb = torch.empty(int(K.lib.lavt_window_attn_bwd_ws(K.dt(bf), nwin, N, heads, ld, wd, ws, ws)), device=dev)
table = torch.randn(R, heads, device=dev) * 0.1
# ATTN_SHIFT=1: the shifted block's region ids (one image of nwin windows; three bands per axis as in the reference's mask)
region = None
if os.environ.get("ATTN_SHIFT", "0") == "1":
    region = torch.randint(0, 3, (nwin, N), device=dev, dtype=torch.int8)
rg, nwi = (K.ptr(region), nwin) if region is not None else (None, 0)
for _ in range(6):
    K.check(K.lib.lavt_window_attn_fwd(K.dt(bf), K.ptr(qkv), None, ld, rg, nwi, K.ptr(out), K.ptr(lse), K.ptr(table), wd, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
    K.check(K.lib.lavt_window_attn_bwd(K.dt(bf), K.ptr(qkv), None, ld, rg, nwi, K.ptr(out), K.ptr(dout), K.ptr(lse), K.ptr(dqkv), K.ptr(table), K.ptr(dtable), K.ptr(wsb), wsb.numel(), None, wd, ws, ws, nwin, N, heads, 32, 32 ** -0.5, K.stream()))
torch.cuda.synchronize()
