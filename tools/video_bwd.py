"""Dev check: backbone-only backward of the micro video model (loss = <w, feature_k>), GPU fp32 vs float64 oracle."""
import os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "lavt-rs_amd")]
from test_gpu_modules import _build_video
from lavt_hip.detweights import det_inputs
from oracle import lavt_video_oracle as OV
tag = sys.argv[1] if len(sys.argv) > 1 else "sept"
which = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
model = _build_video(tag).train()
frames, l, m, tgt = det_inputs(2, 64, 22, seed=123, frames=4)
bb = model["backbone"]
dt = torch.float64
f = bb(frames.cuda().permute(0, 2, 1, 3, 4), l.cuda(), m.cuda())
g = torch.Generator().manual_seed(3)
ws = [torch.randn(*fi.shape, generator=g) for fi in f]
sum((fi * wi.cuda()).sum() for k, (fi, wi) in enumerate(zip(f, ws)) if k in which).backward()
sd = {k[9:]: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("backbone.")}
params = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
full = {"backbone." + k: v for k, v in params.items()}
fo = OV.backbone_3d(full, "backbone", frames.to(dt).permute(0, 2, 1, 3, 4), l.to(dt), m.to(dt), "micro", (8, 7, 7), tag == "sept")
sum((fi * wi.to(dt)).sum() for k, (fi, wi) in enumerate(zip(fo, ws)) if k in which).backward()
agg = collections.defaultdict(float)
for k, p in bb.named_parameters():
    o = params[k].grad
    if p.grad is None or o is None or float(o.norm()) < 1e-6: continue
    e = float((p.grad.cpu().double() - o).abs().max() / o.norm())
    key = ".".join(k.split(".")[:4]) if ("blocks" in k or "fusion" in k) else ".".join(k.split(".")[:2])
    agg[key] = max(agg[key], e)
for k in sorted(agg): print("%-40s %.2e" % (k, agg[k]))
