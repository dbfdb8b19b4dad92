"""Dev check: per-stage forward features of the micro video model, GPU fp32 vs float64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "lavt-rs_amd")]
from test_gpu_modules import _build_video
from lavt_hip.detweights import det_inputs
from oracle import lavt_video_oracle as OV, lavt_oracle as O2
tag = sys.argv[1] if len(sys.argv) > 1 else "sept"
model = _build_video(tag).train()
frames, l, m, tgt = det_inputs(2, 64, 22, seed=123, frames=4)
bb = model["backbone"]
dt = torch.float64
sd = {k: (v.detach().cpu().to(dt) if v.dtype.is_floating_point else v.cpu()) for k, v in model.state_dict().items()}
with torch.no_grad():
    t, T, Wh, Ww = bb.patch_embed.tokens(frames.cuda().permute(0, 2, 1, 3, 4), torch.float32)
    t = t.view(2, T, Wh, Ww, 32)
    vid = frames.to(dt).permute(0, 2, 1, 3, 4)
    y = torch.nn.functional.conv3d(vid, sd["backbone.patch_embed.proj.weight"], sd["backbone.patch_embed.proj.bias"], stride=(1, 4, 4))
    x = O2._ln(y.permute(0, 2, 3, 4, 1), sd, "backbone.patch_embed.norm")
    print("patch embed", float((t.cpu().double() - x).abs().max()))
    for i, layer in enumerate(bb.layers):
        p = f"backbone.layers.{i}"
        xb, tb = x, t
        for b, blk in enumerate(layer.blocks):
            tb = blk(tb); xb = OV.swin_block_3d(sd, f"{p}.blocks.{b}", xb, bb.layers[i].blocks[0].num_heads, (8, 7, 7), shifted=(b % 2 == 1))
            print(f"  stage {i} block {b}: {float((tb.cpu().double() - xb).abs().max()):.2e}  (max |x| {float(xb.abs().max()):.2f})")
        B, D, H, W, C = xb.shape
        from lib.backbone import _LangCtx
        lang = _LangCtx.get(l.cuda(), m.cuda(), torch.float32)
        # fusion on the ORACLE's block output, so errors do not compound
        xin = xb.float().cuda()
        r = layer.fusion.rows(xin.reshape(-1, C), B, D, H, W, lang) if tag == "sept" else layer.fusion.rows(xin.reshape(-1, C), B, D * H * W, lang)
        ro = OV.sep_t_pwam(sd, p + ".fusion", xb, l.to(dt), m.to(dt)) if tag == "sept" else O2.pwam(sd, p + ".fusion", xb.reshape(B, -1, C), l.to(dt), m.to(dt))
        print(f"  stage {i} fusion on exact input: {float((r.cpu().double().view_as(ro) - ro).abs().max()):.2e}  (max |r| {float(ro.abs().max()):.2f})")
        f, t = layer.rows(t, l.cuda(), m.cuda())
        fo, x = OV.stage_3d(sd, p, x, l.to(dt), m.to(dt), len(layer.blocks), layer.blocks[0].num_heads, (8, 7, 7), last=(i == 3), sep_t=(tag == "sept"))
        print(f"stage {i}: feature {float((f.cpu().double() - fo).abs().max()):.2e}  next {float((t.cpu().double() - x).abs().max()):.2e}")
