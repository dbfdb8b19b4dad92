#!/usr/bin/env python3
"""Which torch (at::) kernels and copies are still launched inside one training step, and from which Python lines (torch.profiler with stacks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lavt-rs_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import lavt_hip, bench
from lavt_hip.detweights import det_inputs
from lavt_hip.engine import TrainStep
dev = torch.device("cuda", 0)
lavt_hip.set_compute_dtype("bf16")
cfg = dict(bench.WORKLOADS["swin_b_w12_480_b2"], name="swin_b_w12_480_b2")
model = bench.build_model(cfg, dev, 0.3).train()
x, l, m, t = det_inputs(2, 480, 20, seed=1234)
step = TrainStep(model, x.to(dev), l.to(dev), m.to(dev), t.to(dev), use_graph=False)
step.warmup_and_capture(eager_iters=2)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::reshape", "aten::as_strided", "aten::permute", "aten::detach", "aten::empty_strided", "aten::slice", "aten::select", "aten::alias", "aten::_unsafe_view", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::t", "aten::transpose", "aten::view_as", "aten::contiguous", "aten::_reshape_alias", "aten::to", "aten::lift_fresh", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero"):
        continue
    st = [f for f in (ev.stack or []) if "lavt" in f or "lib/" in f or "bench" in f]
    key = (ev.name, st[0] if st else "?")
    rows[key] = rows.get(key, 0) + 1
for (name, where), n in sorted(rows.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{n:4d}  {name:28s} {where[-110:]}")
