#!/usr/bin/env python3
"""The fused PWAM + language-gate node (reference lib/backbone.py:1265-1278, 1329-1372, 604-611) at the four stage shapes of the headline workload
(Swin-B, 480x480, batch 2: C = 128 ... 1024, T = 14400 ... 225 pixels, 20 words), forward + backward captured in ONE hipGraph and replayed: run under
`rocprofv3 --kernel-trace --output-format csv` and group the trace with tools/trace_by_shape.py <csv> 30 to get per-launch times in replay.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pw -- python3 tools/pwam_node_time.py
"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch  # noqa: E402
import lavt_hip  # noqa: E402
from lavt_hip.detweights import fill_state_dict_  # noqa: E402
from lib.backbone import MMBasicLayer  # noqa: E402

DEV = "cuda:0"
REPLAYS = 30
lavt_hip.set_compute_dtype(torch.bfloat16)
B, n_l = 2, 20
nodes = []
for C, T in ((128, 14400), (256, 3600), (512, 900), (1024, 225)):
    st = MMBasicLayer(dim=C, depth=0, num_heads=C // 32, window_size=12, drop_path=0.0, downsample=None, num_heads_fusion=1, fusion_drop=0.0,
                      args=SimpleNamespace(swin_type="base")).eval()
    fill_state_dict_(st)
    st.to(DEV)
    g = torch.Generator("cpu").manual_seed(C)
    x = torch.randn(B, T, C, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    l = torch.randn(B, 768, n_l, generator=g).to(DEV).requires_grad_(True)
    m = torch.zeros(B, n_l, 1)
    m[0, :13] = 1
    m[1, :7] = 1
    w = torch.randn(B, T, C, generator=g).to(DEV)
    nodes.append((st, x, l, m.to(DEV), w, T))


def step():
    for st, x, l, m, w, T in nodes:
        r, H, W, xg, _, _ = st(x, T, 1, l, m)
        ((r.float() * w).sum() + (xg.float() * w).sum()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s):
        step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gr.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(REPLAYS):
    gr.replay()
e1.record()
torch.cuda.synchronize()
print(f"four PWAM + gate nodes, forward + backward: {e0.elapsed_time(e1) / REPLAYS * 1e3:.1f} us per replay")
