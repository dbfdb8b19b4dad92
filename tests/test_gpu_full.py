"""GPU parity AT THE BENCHMARKED CONFIGURATIONS (BASELINE.json configs 2-4, the shapes bench.py times), against fixtures captured from the real
reference at those sizes (tests/golden/make_golden.py --only-full ...: full_swin_b_480_b2, full_swin_t_480_b8, full_video_{pwam,sept}_t8_384):

* fp32 compute: decoder output and upsampled logits <= 1e-3, argmax mask identical on decisive pixels, I/U and loss equal, EVERY parameter
  gradient against the reference's digest (train-mode BatchNorm, drop_path 0);
* bf16 compute (the dtype of the timed run), same fixtures.  Stated gate: mask IoU >= 0.98 on decisive pixels (|logit1 - logit0| > 0.25 sigma_logit
  in the reference's fp32 run; measured 0.999 on the image and PWAM video configurations), |d loss| <= 2e-2, and -- because these synthetic random-weight networks give a noise-like logit map with many
  near-tie pixels -- every other figure is held against the REFERENCE'S OWN bf16 run of the same configuration (its CPU bf16-autocast
  forward/backward, stored with the fixture as refbf16_*: at Swin-B 2x480x480 that run agrees with its fp32 self on 98.5 % of the pixels, mask
  IoU 0.906, max |dlogit| 0.39 sigma, gradient digests off by 2.5 % median / 5.8 % p90): pixel agreement >= reference's - 0.5 %, overall mask
  IoU >= reference's - 0.02, max |dlogit| <= 1.5 x reference's, gradient-digest error median / p90 <= 2 x reference's (bf16 activations end
  to end here, fp32 residual stream / norms under the reference's autocast).  The achieved numbers
  are printed (pytest -s);
* the step harness itself (hipGraph replay, fused upsample+CE, grouped weight gradients, flat gradient buffer) at the bench shape.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lavt_hip.detweights import det_inputs, fill_state_dict_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IMAGE = {"swin_b": ("full_swin_b_480_b2", 128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 2),
         "swin_b_b4": ("full_swin_b_480_b4", 128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 4),          # BASELINE configs[4]'s batch (4 per GPU)
         "swin_t": ("full_swin_t_480_b8", 96, [2, 2, 6, 2], [3, 6, 12, 24], 7, 8)}
SEPT = dict(sep_t_pwam=True, conv3d_kernel_size_t="3-3-3", conv3d_kernel_size_s="1-1-1", w_t3x3_s1x1=True, mm_t3x3_s1x1=True)
# gradients the bench's hot kernels produce (the grouped stage-2 weight gradients, the 256x256-tile decoder convolutions, ...): always reported
NAMED = ["backbone.layers.2.blocks.17.attn.qkv.weight", "backbone.layers.2.blocks.17.attn.proj.weight", "backbone.layers.2.blocks.17.mlp.fc1.weight",
         "backbone.layers.2.blocks.17.mlp.fc2.weight", "backbone.layers.2.blocks.17.attn.relative_position_bias_table", "backbone.layers.2.blocks.17.norm1.weight",
         "backbone.layers.2.blocks.0.attn.qkv.weight", "backbone.layers.2.blocks.9.mlp.fc1.weight", "backbone.layers.0.blocks.1.attn.qkv.weight",
         "backbone.layers.0.blocks.1.mlp.fc2.weight", "backbone.layers.1.blocks.0.attn.proj.weight", "backbone.layers.3.blocks.1.mlp.fc1.weight",
         "backbone.layers.0.fusion.vis_project.0.weight", "backbone.layers.2.fusion.image_lang_att.f_key.0.weight", "backbone.layers.2.fusion.image_lang_att.W.0.weight",
         "backbone.layers.1.res_gate.0.weight", "backbone.layers.1.downsample.reduction.weight", "backbone.patch_embed.proj.weight", "backbone.norm2.weight",
         "classifier.conv1_4.weight", "classifier.conv2_4.weight", "classifier.conv1_3.weight", "classifier.conv2_3.weight", "classifier.conv1_2.weight",
         "classifier.conv2_2.weight", "classifier.bn2_2.weight", "classifier.conv1_1.weight"]


def grad_digest(t, n=24):
    f = t.detach().reshape(-1).double().cpu()
    step = max(f.numel() // (n // 2), 1)
    samp = torch.cat([f[: n // 2], f[::step][: n // 2]])
    samp = F.pad(samp, (0, n - samp.numel()))
    return torch.cat([torch.stack([f.norm(), f.sum()]), samp]).float()


@pytest.fixture(autouse=True)
def _restore():
    import lavt_hip
    from lavt_hip import ops
    lavt_hip.set_compute_dtype(torch.float32)
    yield
    ops.sinks.clear()
    ops.wgrads.enabled = False
    lavt_hip.set_compute_dtype(torch.float32)
    torch.cuda.empty_cache()


def _image_model(embed, depths, heads, ws):
    from lib._utils import LAVT
    from lib.backbone import MultiModalSwinTransformer
    from lib.mask_predictor import SimpleDecoding
    a = SimpleNamespace()
    model = LAVT(MultiModalSwinTransformer(embed_dim=embed, depths=depths, num_heads=heads, window_size=ws, drop_path_rate=0.0, args=a), SimpleDecoding(8 * embed, a))
    fill_state_dict_(model)
    return model.to(DEV).train()


def _video_model(sept):
    from lib.mask_predictor import SimpleDecoding
    from lib.video_swin_transformer import MultiModalSwinTransformer3D
    a = SimpleNamespace(**(SEPT if sept else {}))
    bb = MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7),
                                     drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1],
                                     fusion_drop=0.0, args=a)
    model = torch.nn.ModuleDict({"backbone": bb, "classifier": SimpleDecoding(1024, a)})
    fill_state_dict_(model)
    return model.to(DEV).train()


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.as_tensor(np.unpackbits(bits)[:n].reshape(shape)).bool()


def _forward(model, video, x, l, m):
    from lib._utils import _upsample_logits
    if video:
        feats = model["backbone"](x.permute(0, 2, 1, 3, 4), l, m)
        lowres = model["classifier"](feats[3], feats[2], feats[1], feats[0])
    else:
        feats = model.backbone(x, l, m)
        lowres = model.classifier(feats[3], feats[2], feats[1], feats[0])
    return feats, lowres, _upsample_logits(lowres, x.shape[-2:])


def _check_forward(g, feats, lowres, logits, tgt, loss, fp32):
    """-> dict of achieved numbers; asserts the gate of the dtype"""
    size = logits.shape[-1]
    lg = logits.detach().float().cpu()
    ref_low = torch.as_tensor(g["lowres"])
    ref_s = torch.as_tensor(g["logits_s"])
    stride = 4 if lg.shape[0] <= 2 else 8
    got_s = lg[:, :, 1::stride, 2::stride]
    sigma = float(g["logit_std"])
    ref_mask = _unpack(g["mask"], (lg.shape[0], size, size))
    decisive = _unpack(g["decisive"], (lg.shape[0], size, size))
    pred = lg.argmax(1).bool()
    r = dict(err_lowres=float((lowres.detach().float().cpu() - ref_low).abs().max()), err_logits=float((got_s - ref_s).abs().max()), sigma=sigma,
             agree_decisive=float((pred[decisive] == ref_mask[decisive]).float().mean()),
             mask_iou=float((pred & ref_mask).sum()) / max(float((pred | ref_mask).sum()), 1.0),
             dloss=abs(float(loss) - float(g["loss"])), ties=int((~decisive).sum()))
    I, U = int((pred & tgt.bool().cpu()).sum()), int((pred | tgt.bool().cpu()).sum())
    r["dI"], r["dU"] = abs(I - int(g["I"])), abs(U - int(g["U"]))
    for i, f in enumerate(feats):
        st = max(f.shape[-1] // 15, 1)
        r[f"err_c{i + 1}"] = float((f.detach().float().cpu()[:, ::8, ::st, ::st] - torch.as_tensor(g[f"c{i + 1}_s"])).abs().max())
    dq = _unpack(g["decisive_q"], (lg.shape[0], size, size))
    r["iou_decisive_q"] = float((pred & ref_mask & dq).sum()) / max(float(((pred | ref_mask) & dq).sum()), 1.0)
    if not fp32:
        r["reference_bf16"] = {k: round(float(g[k]), 5) for k in ("refbf16_agree", "refbf16_iou", "refbf16_maxerr", "refbf16_dloss")}
    print(f"\n[forward {'fp32' if fp32 else 'bf16'}] {r}")
    if fp32:
        assert r["err_lowres"] <= 1e-3 and r["err_logits"] <= 1e-3, r
        assert r["agree_decisive"] == 1.0, r
        assert r["dI"] <= r["ties"] and r["dU"] <= r["ties"], r
        assert r["dloss"] < 1e-4, r
        assert max(r[f"err_c{i}"] for i in range(1, 5)) <= 1e-3, r
    else:
        agree = float((pred == ref_mask).float().mean())
        r["agree"] = agree
        # Gate on the decisive pixels (|margin| > 0.25 sigma): IoU >= 0.98 -- for the configurations whose reference-bf16 run itself keeps logits
        # within 0.5 sigma.  Video-SepTPWAM does not (27-tap convolutions: the reference's own autocast run moves logits by up to 1.1 sigma, its
        # mask IoU is 0.757), so a 0.25-sigma margin is not decisive there and that configuration is gated as "no worse than the reference's own
        # bf16 run" by the agreement / IoU / max-error asserts below (this build: IoU on decisive pixels 0.924, mask IoU 0.770).
        if float(g["refbf16_maxerr"]) <= 0.5 * sigma:
            assert r["iou_decisive_q"] >= 0.98, r
        else:
            r["decisive_gate"] = "not applied: reference-bf16 max error %.2f sigma > 0.5 sigma" % (float(g["refbf16_maxerr"]) / sigma)
        assert r["dloss"] <= 2e-2, r
        assert agree >= float(g["refbf16_agree"]) - 0.005, r
        assert r["mask_iou"] >= float(g["refbf16_iou"]) - 0.02, r
        assert max(r["err_lowres"], r["err_logits"]) <= 1.5 * float(g["refbf16_maxerr"]), r          # a maximum over ~10^6 values: noisy (1.27x seen once)
    return r


def _check_grads(g, named_grads, fp32, tol32=3e-3):
    """named_grads: iterable of (name, grad tensor or None).  Digest = [l2 norm, sum, 12 leading + 12 strided samples] of the reference's gradient.
    fp32: every digest entry (the plain sum aside: cancellation noise) within `tol32` of the gradient's norm, for EVERY parameter.
    bf16: per-parameter error e = max(|norm - ref| / ref, max sample error / ref norm / 1.5); the median and the 90th percentile over the
    parameters must stay within 2 x what the reference's own bf16 run shows (refbf16_grad_*), and no parameter beyond 3 x its worst."""
    nograd = set(g["nograd"].tolist())
    worst, bad, seen = {}, [], 0
    for k, grad in named_grads:
        if k in nograd:
            assert grad is None or float(grad.abs().max()) == 0.0, k
            continue
        ref = torch.as_tensor(g["g|" + k])
        d = grad_digest(grad)
        norm = float(ref[0])
        seen += 1
        if norm <= 1e-6:                 # analytically zero gradients (biases in front of an InstanceNorm): rounding noise on both sides
            assert float(d[0]) <= (1e-4 if fp32 else 2e-3), (k, float(d[0]))
            continue
        if fp32:
            e = float((torch.cat([d[:1], d[2:]]) - torch.cat([ref[:1], ref[2:]])).abs().max()) / norm
            if not e <= tol32 + 5e-6 / norm:
                bad.append((k, round(e, 5), norm))
        else:
            e = max(abs(float(d[0]) - norm) / norm, float((d[2:] - ref[2:]).abs().max()) / norm / 1.5)
        worst[k] = e
    assert seen >= 20
    es = sorted(worst.values())
    out = dict(n=seen, median=round(es[len(es) // 2], 5), p90=round(es[int(len(es) * 0.9)], 5), max=round(es[-1], 5),
               worst=[(k, round(v, 5)) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:3]],
               named={k: round(worst[k], 5) for k in NAMED if k in worst})
    if not fp32:
        out["reference_bf16"] = {k: round(float(g["refbf16_grad_" + k]), 5) for k in ("median", "p90", "max")}
    print(f"\n[gradients {'fp32' if fp32 else 'bf16'}] {out}")
    if fp32:
        assert not bad, f"{len(bad)} of {seen} parameter gradients off: {bad[:10]}"
    else:
        # (the reference's autocast keeps the residual stream, LayerNorm and softmax in fp32; this path keeps bf16 activations end to end, so its
        # gradient noise sits above the reference's bf16 run -- measured 1.3x median / 1.5x p90 at Swin-B 2x480x480 -- and is gated at 2x)
        assert out["median"] <= 2.0 * float(g["refbf16_grad_median"]), out
        assert out["p90"] <= 2.0 * float(g["refbf16_grad_p90"]), out
        assert out["max"] <= 3.0 * max(float(g["refbf16_grad_max"]), 0.1), out
    return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["swin_b", "swin_t", "swin_b_b4"])
def test_full_image_config(golden, tag, dtype):
    """BASELINE configs[2] (Swin-B w12, 2x480x480: 30->36 / 15->24 padded windows at C=512/1024, M=28 800 decoder convolutions on the
    256x256 tile) and configs[1] (Swin-T w7, 8x480x480: 126-row padded grid)."""
    import lavt_hip
    name, embed, depths, heads, ws, B = IMAGE[tag]
    g = golden(name)
    lavt_hip.set_compute_dtype(torch.float32 if dtype == "fp32" else torch.bfloat16)
    model = _image_model(embed, depths, heads, ws)
    x, l, m, tgt = det_inputs(B, 480, 20, seed=int(g["seed"]))
    x, l = x.to(DEV).requires_grad_(True), l.to(DEV).requires_grad_(True)
    feats, lowres, logits = _forward(model, False, x, l, m.to(DEV))
    loss = F.cross_entropy(logits, tgt.to(DEV), weight=torch.tensor([0.9, 1.1], device=DEV))
    r = _check_forward(g, feats, lowres, logits, tgt, loss.detach(), dtype == "fp32")
    loss.backward()
    gr = _check_grads(g, [(k, p.grad) for k, p in model.named_parameters()] , dtype == "fp32")
    for nm, t in (("dx", x.grad), ("dl", l.grad)):
        ref = torch.as_tensor(g[nm])
        e = float((grad_digest(t)[2:] - ref[2:]).abs().max()) / float(ref[0])
        assert e <= (3e-3 if dtype == "fp32" else 0.15), (nm, e)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["pwam", "sept"])
def test_full_video_config(golden, tag, dtype):
    """BASELINE configs[3]: Video-Swin-B, one clip of T=8 frames at 384x384, 22 tokens; PWAM and the README SepTPWAM recipe
    (lib/video_swin_transformer.py:854-881).  392-token windows: fp32 on the composed path, bf16 on the fused 25-tile kernels."""
    import lavt_hip
    g = golden(f"full_video_{tag}_t8_384")
    lavt_hip.set_compute_dtype(torch.float32 if dtype == "fp32" else torch.bfloat16)
    model = _video_model(tag == "sept")
    frames, l, m, tgt = det_inputs(1, 384, 22, seed=int(g["seed"]), frames=8)
    frames, l = frames.to(DEV).requires_grad_(True), l.to(DEV).requires_grad_(True)
    feats, lowres, logits = _forward(model, True, frames, l, m.to(DEV))
    loss = F.cross_entropy(logits, tgt.to(DEV), weight=torch.tensor([0.9, 1.1], device=DEV))
    r = _check_forward(g, feats, lowres, logits, tgt, loss.detach(), dtype == "fp32")
    gr = None
    if not int(g["fwd_only"]):
        loss.backward()
        # SepTPWAM: five 27-tap convolutions + instance norms per stage -- the float32 CPU runs themselves sit ~2e-3 apart (see the micro fixture's test)
        gr = _check_grads(g, [(k, p.grad) for k, p in model.named_parameters()], dtype == "fp32", tol32=6e-3 if tag == "sept" else 3e-3)
    del r, gr


def test_bench_step_matches_reference(golden):
    """What bench.py times -- TrainStep on Swin-B w12, 2x480x480, bf16: hipGraph replay, fused upsample + CE + I/U kernel, grouped weight
    gradients into the flat buffer -- against the reference's fp32 run of the same inputs (drop_path 0): loss, I/U, every gradient."""
    import lavt_hip
    from lavt_hip.engine import TrainStep
    name, embed, depths, heads, ws, B = IMAGE["swin_b"]
    g = golden(name)
    lavt_hip.set_compute_dtype(torch.bfloat16)
    model = _image_model(embed, depths, heads, ws)
    x, l, m, tgt = det_inputs(B, 480, 20, seed=int(g["seed"]))
    step = TrainStep(model, x.to(DEV), l.to(DEV), m.to(DEV), tgt.to(DEV))
    step.warmup_and_capture(eager_iters=2)
    assert step.captured and step.fused_loss
    step.step()
    step.step()
    torch.cuda.synchronize()
    dloss = abs(float(step.loss) - float(g["loss"]))
    stats = step.stats.cpu()
    dI, dU = abs(int(stats[2]) - int(g["I"])), abs(int(stats[3]) - int(g["U"]))
    print(f"\n[bench step bf16 graph] dloss {dloss:.5f} dI {dI} dU {dU} of I {int(g['I'])} U {int(g['U'])}")
    assert dloss <= 2e-2, dloss
    n_pix = B * 480 * 480
    flips = (1.0 - float(g["refbf16_agree"]) + 0.005) * n_pix          # the I / U counts may move by what the reference's own bf16 run flips
    assert dI <= flips and dU <= flips, (dI, dU, flips)
    _check_grads(g, [(k, p.grad) for k, p in model.named_parameters()], False)


@pytest.mark.parametrize("tag", ["swin_b", "swin_b_b4"])
def test_full_swin_b_fp8(golden, tag):
    """BASELINE configs[4] arithmetic (e4m3 weights / activations on the fp8 MFMA for the decoder's 3x3 convolutions -- 54 % of the FLOPs -- bf16
    elsewhere) on the Swin-B 2x480x480 fixture and at configs[4]'s own batch (4x480x480), after one calibration step (delayed scaling).  The reference has no fp8 path: the gate is stated
    against its fp32 run, next to the reference's own bf16 figures -- mask IoU on decisive pixels (|margin| > 0.25 sigma) >= 0.97,
    pixel agreement >= the reference-bf16 agreement - 0.02, |d loss| <= 3e-2, gradient-digest error median / p90 <= 3 x the reference-bf16's."""
    import lavt_hip
    from lavt_hip import ops
    name, embed, depths, heads, ws, B = IMAGE[tag]
    g = golden(name)
    ops.fp8.__init__()
    with lavt_hip.use_dtype("fp8"):
        model = _image_model(embed, depths, heads, ws)
        x, l, m, tgt = det_inputs(B, 480, 20, seed=int(g["seed"]))
        x, l, m = x.to(DEV), l.to(DEV), m.to(DEV)
        with torch.no_grad():                              # calibration pass: records the |max| of every quantisation site
            ops.fp8.advance()
            _forward(model, False, x, l, m)
        ops.fp8.advance()
        feats, lowres, logits = _forward(model, False, x, l, m)
        loss = F.cross_entropy(logits, tgt.to(DEV), weight=torch.tensor([0.9, 1.1], device=DEV))
        loss.backward()
    size = 480
    lg = logits.detach().float().cpu()
    ref_mask, dq = _unpack(g["mask"], (B, size, size)), _unpack(g["decisive_q"], (B, size, size))
    pred = lg.argmax(1).bool()
    r = dict(agree=float((pred == ref_mask).float().mean()), mask_iou=float((pred & ref_mask).sum()) / float((pred | ref_mask).sum()),
             iou_decisive_q=float((pred & ref_mask & dq).sum()) / max(float(((pred | ref_mask) & dq).sum()), 1.0),
             err_lowres=float((lowres.detach().float().cpu() - torch.as_tensor(g["lowres"])).abs().max()), sigma=float(g["logit_std"]),
             dloss=abs(float(loss) - float(g["loss"])), reference_bf16_agree=float(g["refbf16_agree"]), reference_bf16_iou=float(g["refbf16_iou"]))
    print(f"\n[forward fp8] {r}")
    assert r["iou_decisive_q"] >= 0.97 and r["agree"] >= float(g["refbf16_agree"]) - 0.02 and r["dloss"] <= 3e-2, r
    worst = {}
    for k, p in model.named_parameters():
        if p.grad is None or ("g|" + k) not in g.files or float(g["g|" + k][0]) <= 1e-6:
            continue
        ref = torch.as_tensor(g["g|" + k])
        d = grad_digest(p.grad)
        worst[k] = max(abs(float(d[0]) - float(ref[0])) / float(ref[0]), float((d[2:] - ref[2:]).abs().max()) / float(ref[0]) / 1.5)
    es = sorted(worst.values())
    out = dict(n=len(es), median=round(es[len(es) // 2], 5), p90=round(es[int(len(es) * 0.9)], 5), max=round(es[-1], 5),
               reference_bf16={k: round(float(g["refbf16_grad_" + k]), 5) for k in ("median", "p90", "max")})
    print(f"[gradients fp8] {out}")
    assert out["median"] <= 3.0 * float(g["refbf16_grad_median"]) and out["p90"] <= 3.0 * float(g["refbf16_grad_p90"]), out
