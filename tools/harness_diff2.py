#!/usr/bin/env python3
"""First module (in backward order) whose incoming gradient differs between plain autograd and the harness with the fused loss."""
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
lavt_hip.set_compute_dtype(torch.bfloat16)
x, l, m, t = [v.to(DEV) for v in det_inputs(2, 96, 20, seed=3)]
def build():
    md = segmentation.lavt("", SimpleNamespace(swin_type="tiny", drop_path_rate=0.0)); fill_state_dict_(md); return md.to(DEV).train()
def hook_all(md, rec):
    for name, mod in md.named_modules():
        if name and name.count(".") <= 4:
            def fn(mod_, gin, gout, name=name):
                g = [o for o in gout if o is not None]
                if g: rec.append((name, g[0].detach().float().clone()))
            mod.register_full_backward_hook(fn)
recA, recB = [], []
md = build(); hook_all(md, recA)
loss = F.cross_entropy(md(x, l, m), t, weight=torch.tensor([0.9, 1.1], device=DEV)); loss.backward()
md2 = build(); hook_all(md2, recB)
st = TrainStep(md2, x, l, m, t, use_graph=False, fused_loss=os.environ.get("FUSED_LOSS", "1") == "1")
st.warmup_and_capture(eager_iters=1)
recB.clear()
st.step(); torch.cuda.synchronize()
print(len(recA), len(recB))
for (na, ga), (nb, gb) in zip(recA, recB):
    if na != nb or ga.shape != gb.shape:
        print("order differs", na, nb); break
    rel = float((ga - gb).norm() / ga.norm().clamp_min(1e-20))
    if rel > float(os.environ.get("THR", 1e-5)):
        print(f"{rel:.3e}  {na}  {tuple(ga.shape)}")
