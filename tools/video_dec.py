"""Dev check: decoder + upsample + CE on exact (oracle) video features, GPU fp32 vs float64 oracle."""
import os, sys, collections
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "lavt-rs_amd")]
from test_gpu_modules import _build_video
from lavt_hip.detweights import det_inputs
from lib._utils import _upsample_logits
from oracle import lavt_video_oracle as OV, lavt_oracle as O2
tag = sys.argv[1] if len(sys.argv) > 1 else "sept"
model = _build_video(tag).train()
frames, l, m, tgt = det_inputs(2, 64, 22, seed=123, frames=4)
dt = torch.float64
sd = {k: (v.detach().cpu().to(dt) if v.dtype.is_floating_point else v.cpu()) for k, v in model.state_dict().items()}
with torch.no_grad():
    fo = OV.backbone_3d(sd, "backbone", frames.to(dt).permute(0, 2, 1, 3, 4), l.to(dt), m.to(dt), "micro", (8, 7, 7), tag == "sept")
fo = [f.float().double().requires_grad_(True) for f in fo]                # exactly representable in fp32
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("classifier.") and v.dtype.is_floating_point and "running" not in k}
full = dict(sd); full.update(params)
y = O2.decoder(full, "classifier", fo[3], fo[2], fo[1], fo[0], True)
lg = F.interpolate(y, size=(64, 64), mode="bilinear", align_corners=True)
F.cross_entropy(lg, tgt, weight=torch.tensor([0.9, 1.1], dtype=dt)).backward()
fg = [f.detach().float().cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True) for f in fo]
yg = model["classifier"](fg[3], fg[2], fg[1], fg[0])
logits = _upsample_logits(yg, (64, 64))
F.cross_entropy(logits, tgt.cuda(), weight=torch.tensor([0.9, 1.1], device="cuda")).backward()
print("logits", float((logits.detach().cpu().double() - lg).abs().max()))
for i in range(4):
    print("d feature", i, float((fg[i].grad.cpu().double() - fo[i].grad).abs().max() / fo[i].grad.norm()), "mean/std of feature", float(fo[i].mean()), float(fo[i].std()))
for k, p in model["classifier"].named_parameters():
    o = params["classifier." + k].grad
    print("%-20s %.2e" % (k, float((p.grad.cpu().double() - o).abs().max() / o.norm())))
