#!/usr/bin/env python3
"""Which Python lines of one eager training step issue device-to-device copies / torch element-wise kernels (glue launches)?  GPU box only."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import bench, lavt_hip
from lavt_hip.detweights import det_inputs
from lavt_hip.engine import TrainStep
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
cfg = dict(bench.WORKLOADS["swin_b_w12_480_b2"], name="swin_b_w12_480_b2")
lavt_hip.set_compute_dtype("bf16")
model = bench.build_model(cfg, dev).train()
x, l, m, t = det_inputs(2, 480, 20, seed=1234)
step = TrainStep(model, x.to(dev), l.to(dev), m.to(dev), t.to(dev), use_graph=False)
step.warmup_and_capture()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::cat", "aten::_to_copy", "aten::floor", "aten::div", "aten::rand", "aten::sub", "aten::zeros", "aten::contiguous", "aten::sum"):
        st = [s for s in (ev.stack or []) if "lavt" in s or "lib/" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d} x {n:18s} {s}")
