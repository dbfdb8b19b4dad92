"""Dev check: SepTPWAM forward/backward on the GPU (fp32) vs the float64 CPU oracle, per tensor."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "lavt-rs_amd")]
from types import SimpleNamespace
from lavt_hip.detweights import fill_state_dict_
from lib.video_swin_transformer import SepTPWAM
from oracle import lavt_video_oracle as OV
for (B, D, H, W, C) in ((2, 4, 6, 5, 32), (2, 4, 16, 16, 32), (1, 4, 2, 2, 256)):
    sp = SepTPWAM(C, C, 768, C, C, num_heads=1, conv3d_kernel_size_t=(3, 3, 3), conv3d_kernel_size_s=(1, 1, 1), w_t3x3_s1x1=True, mm_t3x3_s1x1=True, args=SimpleNamespace())
    fill_state_dict_(sp); sp.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, H, W, C, generator=g); l = torch.randn(B, 768, 20, generator=g); w = torch.randn(B, D * H * W, C, generator=g)
    m = torch.zeros(B, 20, 1); m[0, :6] = 1; m[-1, :17] = 1
    xg = x.cuda().requires_grad_(True); lg = l.cuda().requires_grad_(True)
    y = sp(xg, lg, m.cuda()); (y * w.cuda()).sum().backward()
    sd = {"f." + k: v.detach().cpu().double().requires_grad_(True) for k, v in sp.state_dict().items()}
    xo = x.double().requires_grad_(True); lo = l.double().requires_grad_(True)
    yo = OV.sep_t_pwam(sd, "f", xo, lo, m.double()); (yo * w.double()).sum().backward()
    print((B, D, H, W, C), "fwd %.2e" % float((y.cpu().double() - yo).abs().max() / yo.abs().max()))
    def rel(a, b): return float((a.cpu().double() - b).abs().max() / (b.norm() + 1e-30))
    print("   dx %.2e  dl %.2e" % (rel(xg.grad, xo.grad), rel(lg.grad, lo.grad)))
    for k, p in sp.named_parameters():
        o = sd["f." + k].grad
        if float(o.norm()) > 1e-6: print("   %-36s %.2e" % (k, rel(p.grad, o)))
