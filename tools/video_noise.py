"""Dev check: gradient noise of the micro video model (SepTPWAM) -- GPU fp32 vs float64 CPU oracle vs the reference's float32 digests."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "lavt-rs_amd")]
from test_gpu_modules import _build_video, grad_digest
from lavt_hip.detweights import det_inputs
from lib._utils import _upsample_logits
from oracle import lavt_video_oracle as OV
tag = sys.argv[1] if len(sys.argv) > 1 else "sept"
g = np.load(os.path.join(ROOT, "tests/golden", f"e2e_video_micro_{tag}.npz"))
model = _build_video(tag).train()
frames, l, m, tgt = det_inputs(2, 64, 22, seed=int(g["seed"]), frames=4)
f = model["backbone"](frames.cuda().permute(0, 2, 1, 3, 4), l.cuda(), m.cuda())
logits = _upsample_logits(model["classifier"](f[3], f[2], f[1], f[0]), frames.shape[-2:])
F.cross_entropy(logits, tgt.cuda(), weight=torch.tensor([0.9, 1.1], device="cuda")).backward()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
dt = torch.float64
params = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
full = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}; full.update(params)
lg = OV.lavt_video_forward(full, frames.to(dt), l.to(dt), m.to(dt), "micro", (8, 7, 7), sep_t=(tag == "sept"), training=True)
F.cross_entropy(lg, tgt, weight=torch.tensor([0.9, 1.1], dtype=dt)).backward()
rows = []
for k, p in model.named_parameters():
    if p.grad is None or params[k].grad is None: continue
    o = params[k].grad; rms = float(o.norm() / o.numel() ** 0.5); nrm = float(o.norm())
    e_gpu = float((p.grad.cpu().double() - o).abs().max())
    d = grad_digest(o.float()); ref = torch.as_tensor(g["g|" + k])
    e_ref = float((torch.cat([d[:1], d[2:]]) - torch.cat([ref[:1], ref[2:]])).abs().max())
    rows.append((e_gpu / max(nrm, 1e-9), e_ref / max(nrm, 1e-9), k, nrm))
rows = [r for r in rows if r[3] > 1e-6]
rows.sort(reverse=True)
for r in rows[:12]: print("gpu-vs-f64 %.2e   ref32-vs-f64(digest) %.2e   %s  |g|=%.3g" % r)
import collections
agg = collections.defaultdict(float)
for e, _, k, _n in rows:
    key = ".".join(k.split(".")[:4]) if "blocks" in k else ".".join(k.split(".")[:3])
    agg[key] = max(agg[key], e)
for k in sorted(agg): print("%-40s %.2e" % (k, agg[k]))
