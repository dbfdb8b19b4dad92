#!/usr/bin/env python3
"""Uninitialised-read detector: every torch.empty / empty_like CUDA float tensor is filled with NaN (ints with a large value) before the kernels see
it; a forward + backward of the model must still produce finite numbers everywhere.  MODEL=tiny|base, SIZE, HARNESS=1 runs the step harness eagerly."""
import os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch, torch.nn.functional as F
import lavt_hip
from lavt_hip import ops
from lavt_hip.engine import TrainStep
from lavt_hip.detweights import det_inputs, fill_state_dict_
from lib import segmentation
DEV = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float32)
size = int(os.environ.get("SIZE", 96))
x, l, m, t = [v.to(DEV) for v in det_inputs(2, size, 20, seed=3)]
md = segmentation.lavt("", SimpleNamespace(swin_type=os.environ.get("MODEL", "tiny"), drop_path_rate=0.0, window12=os.environ.get("MODEL", "tiny") == "base")); fill_state_dict_(md); md = md.to(DEV).train()
_e, _el = torch.empty, torch.empty_like
def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point(): t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64): t.fill_(1 << 28)
    return t
torch.empty = lambda *a, **k: _poison(_e(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_el(*a, **k))
if os.environ.get("HARNESS", "0") == "1":
    st = TrainStep(md, x, l, m, t, use_graph=False)
    st.warmup_and_capture(eager_iters=1)
    st.step(); torch.cuda.synchronize()
    loss = st.loss
else:
    loss = F.cross_entropy(md(x, l, m), t, weight=torch.tensor([0.9, 1.1], device=DEV)); loss.backward()
torch.cuda.synchronize()
print("loss", float(loss.detach()))
bad = [n for n, p in md.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
print(len(bad), "parameters with non-finite gradients", bad[:12])
