#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by importing the REAL reference.

Runs only in the build container (needs /root/reference).  Nothing of the reference is
copied: this script imports it in-process behind sys.modules stand-ins for its missing
third-party imports (timm, mmcv, mmseg, torchvision, bert -- SURVEY.md 8c / Appendix C),
drives it with deterministic weights (lavt_hip.detweights, keyed by state-dict name) and
seeded inputs, and stores ONLY inputs-by-seed + expected outputs as .npz.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
from lavt_hip.detweights import det_inputs, det_tensor, fill_state_dict_  # noqa: E402
sys.path.remove(os.path.join(ROOT, "lavt-rs_amd"))     # from here on `lib` must resolve to the REFERENCE (a namespace package loses to ours)

REF = "/root/reference"


# ----------------------------------------------------------------------------- shims
def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1 - self.p
            mask = torch.floor(keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1)))
            return x / keep * mask

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=nn.init.trunc_normal_)
    import logging
    mod("mmseg")
    mod("mmseg.utils", get_root_logger=lambda *a, **k: logging.getLogger("ref"))
    mod("mmcv")
    mod("mmcv.fileio", FileClient=object, load=lambda *a, **k: None)
    mod("mmcv.parallel", is_module_wrapper=lambda m: False)
    mod("mmcv.utils", mkdir_or_exist=lambda *a, **k: None)
    mod("mmcv.runner", get_dist_info=lambda: (0, 1))
    tv = mod("torchvision")
    tv.__path__ = []
    mod("torchvision.ops")
    mod("torchvision.ops.boxes", box_area=None)

    class BertModel(nn.Module):
        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

    mod("bert")
    mod("bert.modeling_bert", BertModel=BertModel)


def ref_args(*flags):
    sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, ["x"]
    try:
        import args as ref_args_mod
        return ref_args_mod.get_parser().parse_args(list(flags))
    finally:
        sys.argv = argv


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator("cpu").manual_seed(seed))


def grad_digest(t, n=24):
    """Compact pin of a gradient tensor: [l2, sum, first n/2, strided n/2]."""
    f = t.detach().reshape(-1).double()
    step = max(f.numel() // (n // 2), 1)
    samp = torch.cat([f[: n // 2], f[::step][: n // 2]])
    samp = F.pad(samp, (0, n - samp.numel()))
    return torch.cat([torch.stack([f.norm(), f.sum()]), samp]).float()


# ----------------------------------------------------------------------------- cases
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-video", action="store_true")
    ap.add_argument("--only-checkpoint", action="store_true")
    ap.add_argument("--only-bert", action="store_true")
    ap.add_argument("--only-dice", action="store_true")
    ap.add_argument("--only-w12", action="store_true")
    ap.add_argument("--only-feats", action="store_true", help="LAVTVideo.forward_feats (lib/_utils.py:110-131) on the micro video model")
    ap.add_argument("--only-full", default="", help="comma list of full-size (BASELINE.json configs) cases: swin_b,swin_t,video_pwam,video_sept")
    cli = ap.parse_args()
    if cli.only_bert:
        bert_cases()
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_shims()
    sys.path.insert(0, REF)
    args = ref_args()
    from lib import backbone as rb
    from lib import mask_predictor as rmp
    from lib import _utils as ru
    if cli.only_video:
        video_cases(args)
        return
    if cli.only_full:
        full_cases(args, cli.only_full.split(","))
        return
    if cli.only_dice:
        dice_cases()
        return
    if cli.only_feats:
        forward_feats_case()
        return
    if cli.only_w12:
        video_cases(args, only_w12=True)
        return
    if cli.only_checkpoint:
        checkpoint_cases()
        return

    # --- window attention, with/without mask, both window sizes ------------------------------
    for tag, C, nH, ws, Bw, Hp in (("w7", 96, 3, 7, 8, 14), ("w12", 128, 4, 12, 4, 24)):
        m = rb.WindowAttention(C, (ws, ws), nH).eval()
        fill_state_dict_(m)
        sd = {f"attn.{k}": v for k, v in m.state_dict().items()}
        x = randn(11, Bw, ws * ws, C)
        layer = types.SimpleNamespace(window_size=ws, shift_size=ws // 2)
        mask = _ref_mask(rb, Hp, Hp, ws)
        with torch.no_grad():
            y0 = m(x, None)
            y1 = m(x, mask)
        save(f"win_attn_{tag}", C=C, nH=nH, ws=ws, Bw=Bw, Hp=Hp, seed=11, y_nomask=y0, y_mask=y1)

    # --- shift masks (bit-packed) --------------------------------------------------------------
    packs = {}
    for Hp, ws in ((126, 7), (36, 12), (24, 12), (7, 7), (14, 7)):
        mk = _ref_mask(rb, Hp, Hp, ws)
        assert set(mk.unique().tolist()) <= {0.0, -100.0}
        packs[f"m_{Hp}_{ws}"] = np.packbits((mk != 0).numpy().reshape(-1))
        packs[f"n_{Hp}_{ws}"] = np.array(mk.shape)
    save("shift_masks", **packs)

    # --- Swin blocks (pad 15->24 with ws 12; 28 with ws 7), shifted and not --------------------
    for tag, C, nH, ws, H, B in (("15_w12", 128, 4, 12, 15, 2), ("28_w7", 96, 3, 7, 28, 1), ("10_w7", 64, 2, 7, 10, 2)):
        for shifted in (0, 1):
            blk = rb.SwinTransformerBlock(C, nH, ws, shift_size=(ws // 2 if shifted else 0)).eval()
            fill_state_dict_(blk)
            blk.H = blk.W = H
            x = randn(21, B, H * H, C)
            Hp = -(-H // ws) * ws
            with torch.no_grad():
                y = blk(x, _ref_mask(rb, Hp, Hp, ws))
            save(f"block_{tag}_s{shifted}", C=C, nH=nH, ws=ws, H=H, B=B, seed=21, shifted=shifted, y=y)

    # --- patch merging (even, odd), patch embed (needs padding) ---------------------------------
    for tag, H, W in (("even", 8, 6), ("odd", 7, 5)):
        pm = rb.PatchMerging(32).eval()
        fill_state_dict_(pm)
        x = randn(31, 2, H * W, 32)
        with torch.no_grad():
            y = pm(x, H, W)
        save(f"patch_merging_{tag}", H=H, W=W, C=32, B=2, seed=31, y=y)
    pe = rb.PatchEmbed(4, 3, 48, nn.LayerNorm).eval()
    fill_state_dict_(pe)
    x = randn(32, 2, 3, 30, 27)
    with torch.no_grad():
        y = pe(x)
    save("patch_embed", C=48, seed=32, shape=np.array(x.shape), y=y)

    # --- PWAM with ragged mask, 1 and 2 heads ------------------------------------------------------
    for G in (1, 2):
        pw = rb.PWAM(64, 64, 768, 64, 64, num_heads=G, dropout=0.0).eval()
        fill_state_dict_(pw)
        x = randn(41, 2, 90, 64)
        l = randn(42, 2, 768, 20)
        lm = torch.zeros(2, 20, 1)
        lm[0, :9] = 1
        lm[1, :15] = 1
        with torch.no_grad():
            y = pw(x, l, lm)
            lang = pw.image_lang_att(x, l, lm)
        save(f"pwam_g{G}", C=64, T=90, G=G, seeds=np.array([41, 42]), valid=np.array([9, 15]), y=y, lang=lang)

    # --- one full stage (blocks + PWAM + gate + merge) ---------------------------------------------
    st = rb.MMBasicLayer(dim=64, depth=2, num_heads=2, window_size=7, drop_path=0.0, downsample=rb.PatchMerging,
                         num_heads_fusion=1, fusion_drop=0.0, args=args).eval()
    fill_state_dict_(st)
    x = randn(51, 2, 10 * 9, 64)
    l = randn(52, 2, 768, 20)
    lm = torch.zeros(2, 20, 1)
    lm[0, :7] = 1
    lm[1, :20] = 1
    with torch.no_grad():
        r, H, W, xd, Wh, Ww = st(x, 10, 9, l, lm)
    save("stage_10x9", seeds=np.array([51, 52]), valid=np.array([7, 20]), r=r, x_down=xd, hw=np.array([H, W, Wh, Ww]))

    # --- decoder, eval BN and train-mode BN ------------------------------------------------------
    dec = rmp.SimpleDecoding(64, args)
    fill_state_dict_(dec)
    feats = [randn(60 + i, 2, c, s, s) for i, (c, s) in enumerate(((64, 4), (32, 8), (16, 16), (8, 32)))]
    with torch.no_grad():
        y_eval = dec.eval()(*feats)
        y_train = dec.train()(*feats)
    save("decoder_c64", seeds=np.arange(60, 64), y_eval=y_eval, y_train=y_train)

    # --- end to end --------------------------------------------------------------------------------
    def build(embed_dim, depths, heads, ws, dpr=0.3):
        bb = rb.MultiModalSwinTransformer(embed_dim=embed_dim, depths=depths, num_heads=heads, window_size=ws,
                                          ape=False, drop_path_rate=dpr, patch_norm=True, out_indices=(0, 1, 2, 3),
                                          use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=args)
        bb.init_weights()
        model = ru.LAVT(bb, rmp.SimpleDecoding(8 * embed_dim, args))
        fill_state_dict_(model)
        return model

    # config 1: Swin-T, 1x224x224, 20 tokens (12 valid)
    model = build(96, [2, 2, 6, 2], [3, 6, 12, 24], 7).eval()
    keys = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in model.state_dict().items())
    with open(os.path.join(HERE, "state_dict_keys_swin_t.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")
    x, l, lm, tgt = det_inputs(1, 224, 20, seed=1234)
    lm = torch.zeros(1, 20, 1)
    lm[0, :12] = 1
    with torch.no_grad():
        feats = model.backbone(x, l, lm)
        logits = model(x, l, lm)
    loss = F.cross_entropy(logits, tgt, weight=torch.tensor([0.9, 1.1]))
    pred = logits.argmax(1)
    save("e2e_swin_t_224", seed=1234, valid=12,
         c1=feats[0][:, :, ::4, ::4], c2=feats[1][:, :, ::2, ::2], c3=feats[2], c4=feats[3],
         feat_sums=np.array([float(f.double().sum()) for f in feats]),
         feat_abs=np.array([float(f.double().abs().sum()) for f in feats]),
         logits=logits, mask=np.packbits(pred.numpy().astype(np.uint8).reshape(-1)),
         I=int((pred & tgt).sum()), U=int((pred | tgt).sum()), loss=float(loss),
         margin_frac=float(((logits[:, 1] - logits[:, 0]).abs() > 1e-2).float().mean()))
    print("   swin_t margin>1e-2 fraction:", float(((logits[:, 1] - logits[:, 0]).abs() > 1e-2).float().mean()),
          "logit std", float(logits.std()), "pos frac", float(pred.float().mean()))

    # Swin-B window 12 at a small image: exercises padded windows (24->24, 12, 6->12, 3->12)
    model = build(128, [2, 2, 18, 2], [4, 8, 16, 32], 12).eval()
    keys = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in model.state_dict().items())
    with open(os.path.join(HERE, "state_dict_keys_swin_b_w12.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")
    x, l, lm, tgt = det_inputs(2, 96, 20, seed=77)
    with torch.no_grad():
        logits = model(x, l, lm)
    pred = logits.argmax(1)
    save("e2e_swin_b_w12_96", seed=77, logits=logits, I=int((pred & tgt).sum()), U=int((pred | tgt).sum()))
    print("   swin_b margin>1e-2 fraction:", float(((logits[:, 1] - logits[:, 0]).abs() > 1e-2).float().mean()))

    # tiny model: forward + every parameter gradient (train mode: BN batch stats, drop path 0)
    model = build(32, [2, 2, 2, 2], [1, 2, 4, 8], 7, dpr=0.0).train()
    x, l, lm, tgt = det_inputs(2, 64, 20, seed=99)
    x.requires_grad_(True)
    l.requires_grad_(True)
    logits = model(x, l, lm)
    loss = F.cross_entropy(logits, tgt, weight=torch.tensor([0.9, 1.1]))
    loss.backward()
    digests = {}
    nograd = []
    for k, p in model.named_parameters():
        if p.grad is None:
            nograd.append(k)
        else:
            digests["g|" + k] = grad_digest(p.grad)
    save("e2e_tiny_train", seed=99, logits=logits, loss=float(loss), dx=grad_digest(x.grad), dl=grad_digest(l.grad),
         nograd=np.array(nograd), **digests)
    print("   tiny: params without grad:", nograd)
    video_cases(args)
    checkpoint_cases()
    dice_cases()


def dice_cases():
    """`--loss mc_dice` (train.py:703-704): the reference's losses.MultiClassDiceLoss after the final upsample of lib/_utils.py:21, on seeded
    low-resolution logits and targets (one sample has no foreground pixel at all: its class-1 intersection and target count are 0)."""
    import losses as rl
    crit = rl.MultiClassDiceLoss()
    for tag, (B, h, w, H, W) in {"a": (3, 13, 11, 52, 44), "b": (2, 30, 30, 120, 120), "same": (2, 9, 7, 9, 7)}.items():
        y = (randn(91, B, 2, h, w) * 2.0).requires_grad_(True)
        tgt = (randn(92, B, H, W) > 0.3).long()
        tgt[B - 1] = 0
        up = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=True)
        loss = crit(up, tgt)
        loss.backward()
        save(f"dice_{tag}", dims=np.array([B, h, w, H, W]), seeds=np.array([91, 92]), loss=float(loss), dy=y.grad)


def video_cases(args_base, only_w12=False):
    """Golden vectors of the video path (lib/video_swin_transformer.py), driven through the reference's own classes."""
    import lib.video_swin_transformer as rv
    from lib import mask_predictor as rmp
    rv.sr_ratio = [1, 1, 1, 1]                    # the shipped file reads an undefined global; set from outside

    # --- `--window12` video windows: (8, 12, 12) = 1152 tokens per window (lib/segmentation.py:173-176), un-shifted and shifted (0, 6, 6) ---------
    for shifted in (0, 1):
        window = (8, 12, 12)
        B, D, H, W = 1, 8, 24, 24
        blk = rv.SwinTransformerBlock3D(64, 2, window, shift_size=tuple(w // 2 for w in window) if shifted else (0, 0, 0)).eval()
        fill_state_dict_(blk)
        x = randn(73, B, D, H, W, 64)
        win, shift = rv.get_window_size((D, H, W), window, tuple(w // 2 for w in window))
        mask = rv.compute_mask(D, H, W, win, shift, "cpu")
        with torch.no_grad():
            y = blk(x, mask)
        save(f"vblock_w12_s{shifted}", dims=np.array([B, D, H, W]), seed=73, y=y[:, :, ::2, ::2], ysum=float(y.double().sum()), yabs=float(y.double().abs().sum()))

    if only_w12:
        return

    # --- 3-D shift masks ---------------------------------------------------------------------------------------
    packs = {}
    for tag, (Dp, Hp, Wp, win, shift) in {"a": (8, 14, 14, (8, 7, 7), (0, 3, 3)), "b": (16, 14, 14, (8, 7, 7), (4, 3, 3)),
                                          "c": (16, 7, 7, (8, 7, 7), (4, 0, 0)), "d": (4, 12, 24, (4, 12, 12), (0, 6, 6))}.items():
        mk = rv.compute_mask(Dp, Hp, Wp, win, shift, "cpu")
        packs["m_" + tag] = np.packbits((mk != 0).numpy().reshape(-1))
        packs["n_" + tag] = np.array(mk.shape)
        packs["cfg_" + tag] = np.array([Dp, Hp, Wp, *win, *shift])
    save("video_masks", **packs)

    # --- Video-Swin blocks: full window, clipped window (T=3: index-slice quirk), temporal shift (T=16) ------------------
    for tag, (B, D, H, W) in {"t8": (1, 8, 10, 9), "t3": (1, 3, 10, 9), "t16": (1, 16, 7, 7)}.items():
        for shifted in (0, 1):
            window = (8, 7, 7)
            blk = rv.SwinTransformerBlock3D(64, 2, window, shift_size=tuple(w // 2 for w in window) if shifted else (0, 0, 0)).eval()
            fill_state_dict_(blk)
            x = randn(71, B, D, H, W, 64)
            win, shift = rv.get_window_size((D, H, W), window, tuple(w // 2 for w in window))
            Dp, Hp, Wp = (int(np.ceil(n / w)) * w for n, w in zip((D, H, W), win))
            mask = rv.compute_mask(Dp, Hp, Wp, win, shift, "cpu")
            with torch.no_grad():
                y = blk(x, mask)
            save(f"vblock_{tag}_s{shifted}", dims=np.array([B, D, H, W]), seed=71, y=y)

    # --- SepTPWAM, README recipe -----------------------------------------------------------------------------------
    va = ref_args("--sep_t_pwam", "--conv3d_kernel_size_t", "3-3-3", "--conv3d_kernel_size_s", "1-1-1", "--w_t3x3_s1x1", "--mm_t3x3_s1x1")
    sp = rv.SepTPWAM(32, 32, 768, 32, 32, num_heads=1, dropout=0.0, conv3d_kernel_size_t=(3, 3, 3), conv3d_kernel_size_s=(1, 1, 1),
                     w_t3x3_s1x1=True, mm_t3x3_s1x1=True, args=va).eval()
    fill_state_dict_(sp)
    x, l = randn(81, 2, 4, 6, 5, 32), randn(82, 2, 768, 20)
    lm = torch.zeros(2, 20, 1)
    lm[0, :6] = 1
    lm[1, :17] = 1
    with torch.no_grad():
        y = sp(x, l, lm)
    save("sep_t_pwam", seeds=np.array([81, 82]), valid=np.array([6, 17]), y=y)

    # --- end to end micro video models (PWAM default, SepTPWAM recipe): forward + every parameter gradient ---------------
    for tag, a in (("pwam", ref_args()), ("sept", va)):
        bb = rv.MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                                            window_size=(8, 7, 7), drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3),
                                            use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
        bb.init_weights()
        dec = rmp.SimpleDecoding(256, a)
        model = nn.ModuleDict({"backbone": bb, "classifier": dec})
        fill_state_dict_(model)
        keys = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in model.state_dict().items())
        with open(os.path.join(HERE, f"state_dict_keys_video_micro_{tag}.txt"), "w") as f:
            f.write("\n".join(keys) + "\n")
        model.train()
        # batch 2 on purpose: with batch 1 PyTorch's CPU instance_norm backward mis-reads a channels-last-strided grad_output
        # (the permute+view after f_query_t/f_query_s hands it one), so a batch-1 CPU run of the reference does not give the
        # gradient of its own forward; batch 2 does (checked against float64 and a hand-written normalisation).
        frames, l, lm, tgt = det_inputs(2, 64, 22, seed=123, frames=4)
        frames.requires_grad_(True)
        l.requires_grad_(True)
        feats = bb(frames.permute(0, 2, 1, 3, 4), l, lm)                      # what _LAVTVideoSimpleDecode.forward does (lib/_utils.py:97-106)
        logits = F.interpolate(dec(feats[3], feats[2], feats[1], feats[0]), size=frames.shape[-2:], mode="bilinear", align_corners=True)
        loss = F.cross_entropy(logits, tgt, weight=torch.tensor([0.9, 1.1]))
        loss.backward()
        digests, nograd = {}, []
        for k, p_ in model.named_parameters():
            if p_.grad is None:
                nograd.append(k)
            else:
                digests["g|" + k] = grad_digest(p_.grad)
        save(f"e2e_video_micro_{tag}", seed=123, logits=logits, loss=float(loss.detach()), dframes=grad_digest(frames.grad), dl=grad_digest(l.grad),
             nograd=np.array(nograd), **digests)
        print(f"   video micro {tag}: params without grad: {nograd}")

    # --- state-dict key list of the Video-Swin-B model (factory shapes) ------------------------------------------------------
    for tag, a in (("pwam", ref_args("--swin_type", "base")), ("sept", ref_args("--swin_type", "base", "--sep_t_pwam", "--conv3d_kernel_size_t", "3-3-3",
                                                                                   "--conv3d_kernel_size_s", "1-1-1", "--w_t3x3_s1x1", "--mm_t3x3_s1x1"))):
        bb = rv.MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7),
                                            drop_path_rate=0.3, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False,
                                            num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
        keys = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in bb.state_dict().items())
        with open(os.path.join(HERE, f"state_dict_keys_video_swin_b_{tag}.txt"), "w") as f:
            f.write("\n".join(keys) + "\n")


def forward_feats_case():
    """_LAVTVideoSimpleDecode.forward_feats (reference lib/_utils.py:110-131) through the reference's own class: the text encoder is the shim's stub
    (the reference ships no ./bert) returning fixed language features, everything after it -- permute, Video-Swin backbone, SimpleDecoding.forward_feats
    (lib/mask_predictor.py:102-146), bilinear upsample -- is the reference's code.  Eval mode (running BatchNorm statistics), micro model."""
    import lib.video_swin_transformer as rv
    from lib import _utils as ru
    from lib import mask_predictor as rmp
    rv.sr_ratio = [1, 1, 1, 1]
    a = ref_args()
    bb = rv.MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7),
                                        drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False,
                                        num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
    dec = rmp.SimpleDecoding(256, a)
    frames, l, lm, _ = det_inputs(2, 64, 22, seed=321, frames=4)

    class _Text(nn.Module):                      # stands where BertModel stands: (ids, attention_mask) -> ((B, N_l, 768),)
        def forward(self, ids, attention_mask=None):
            return (l.permute(0, 2, 1),)

    model = ru._LAVTVideoSimpleDecode.__new__(ru._LAVTVideoSimpleDecode)
    nn.Module.__init__(model)
    model.backbone, model.classifier, model.text_encoder = bb, dec, _Text()
    model.lazy_pred, model.seg_last = False, False
    fill_state_dict_(nn.ModuleDict({"backbone": bb, "classifier": dec}))
    model.eval()
    ids = torch.zeros(2, 22, dtype=torch.long)
    with torch.no_grad():
        y, feats = model.forward_feats(frames, ids, lm.squeeze(-1))
    save("video_forward_feats", seed=321, logits=y, **{f"feat{i}": f for i, f in enumerate(feats)}, nfeats=len(feats))
    print("   forward_feats:", tuple(y.shape), [tuple(f.shape) for f in feats])


def _full_record(name, model_params, logits, lowres, feats, tgt, loss, inputs_with_grad, extra):
    """What a full-size fixture holds (inputs are regenerated from the seed, weights from their names): the decoder output before the
    upsample, a strided sample of the upsampled logits, the bit-packed argmax mask and the bit-packed set of decisive pixels
    (|logit1 - logit0| > 2e-3), loss, I/U, per-feature statistics + strided samples, and a digest of every parameter gradient."""
    pred = logits.argmax(1)
    margin = (logits[:, 1] - logits[:, 0]).abs()
    rec = dict(lowres=lowres, logits_s=logits[:, :, 1::4, 2::4] if logits.shape[0] <= 2 else logits[:, :, 1::8, 2::8],
               mask=np.packbits(pred.numpy().astype(np.uint8).reshape(-1)), decisive=np.packbits((margin > 2e-3).numpy().reshape(-1)),
               decisive_q=np.packbits((margin > 0.25 * float(logits.std())).numpy().reshape(-1)),
               I=int((pred & tgt).sum()), U=int((pred | tgt).sum()), loss=float(loss), logit_std=float(logits.std()),
               margin_frac=float((margin > 2e-3).float().mean()),
               feat_sums=np.array([float(f.double().sum()) for f in feats]), feat_abs=np.array([float(f.double().abs().sum()) for f in feats]))
    for i, f in enumerate(feats):
        st = max(f.shape[-1] // 15, 1)
        rec[f"c{i + 1}_s"] = f[:, ::8, ::st, ::st]
    nograd = []
    for k, gr in model_params:                       # (name, gradient tensor or None)
        if gr is None:
            nograd.append(k)
        else:
            rec["g|" + k] = grad_digest(gr)
    rec["nograd"] = np.array(nograd)
    for k, t in inputs_with_grad.items():            # name -> gradient tensor
        rec[k] = grad_digest(t)
    rec.update(extra)
    save(name, **rec)
    print(f"   {name}: loss {float(loss):.6f} logit std {float(logits.std()):.4f} decisive {rec['margin_frac']:.5f} pos {float(pred.float().mean()):.4f} nograd {nograd}")


class _ContigGrad(torch.autograd.Function):
    """identity whose backward hands a CONTIGUOUS gradient upstream: PyTorch's CPU instance_norm backward mis-reads a channels-last-strided
    grad_output at batch 1 (see video_cases); wrapped around F.instance_norm for the batch-1 full-size SepTPWAM run.  The reference's own
    arithmetic is untouched -- this only keeps PyTorch from returning a gradient that is not the gradient of the forward it ran."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _ref_bf16_noise(run, fp32_logits, fp32_loss, fp32_grads):
    """The REFERENCE's own bf16 noise at this configuration: its forward/backward under torch.autocast('cpu', bfloat16) against its fp32
    run.  Stored with the fixture as the yardstick of the bf16 gate (a bf16 implementation cannot be asked to sit closer to the fp32
    reference than the reference's own bf16 run does)."""
    logits, loss, grads = run(True)
    pred, ref = logits.argmax(1).bool(), fp32_logits.argmax(1).bool()
    errs = []
    for k, gr in grads.items():
        ref_d, d = grad_digest(fp32_grads[k]), grad_digest(gr.float())
        norm = float(ref_d[0])
        if norm > 1e-6:
            errs.append(max(abs(float(d[0]) - norm) / norm, float((d[2:] - ref_d[2:]).abs().max()) / norm / 1.5))
    errs = np.sort(np.array(errs)) if errs else np.zeros(1)
    out = dict(refbf16_agree=float((pred == ref).float().mean()), refbf16_iou=float((pred & ref).sum()) / max(float((pred | ref).sum()), 1.0),
               refbf16_maxerr=float((logits - fp32_logits).abs().max()), refbf16_dloss=abs(float(loss) - float(fp32_loss)),
               refbf16_grad_median=float(errs[len(errs) // 2]), refbf16_grad_p90=float(errs[int(len(errs) * 0.9)]), refbf16_grad_max=float(errs[-1]))
    print("   reference's own bf16-autocast noise:", {k: round(v, 5) for k, v in out.items()})
    return out


def full_cases(args, which):
    """BASELINE.json configs at their real sizes (the configurations bench.py times), train mode (batch-statistics BatchNorm), drop_path 0
    (DropPath's random draws cannot be matched across implementations), forward + weighted CE + backward through the reference's classes."""
    import time
    from lib import backbone as rb
    from lib import mask_predictor as rmp
    from lib import _utils as ru
    w = torch.tensor([0.9, 1.1])
    for tag, (embed, depths, heads, ws, B) in {"swin_b": (128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 2), "swin_t": (96, [2, 2, 6, 2], [3, 6, 12, 24], 7, 8),
                                               "swin_b_b4": (128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 4)}.items():          # b4: the per-GPU batch of BASELINE configs[4] (fp8)
        if tag not in which:
            continue
        t0 = time.time()
        bb = rb.MultiModalSwinTransformer(embed_dim=embed, depths=depths, num_heads=heads, window_size=ws, ape=False, drop_path_rate=0.0, patch_norm=True,
                                          out_indices=(0, 1, 2, 3), use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=args)
        model = ru.LAVT(bb, rmp.SimpleDecoding(8 * embed, args))
        fill_state_dict_(model)
        model.train()
        x, l, lm, tgt = det_inputs(B, 480, 20, seed=1234)
        x.requires_grad_(True)
        l.requires_grad_(True)

        def run(autocast):
            model.zero_grad(set_to_none=True)
            x.grad = l.grad = None
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                feats = model.backbone(x, l, lm)
                lowres = model.classifier(feats[3], feats[2], feats[1], feats[0])
            logits = F.interpolate(lowres.float(), size=(480, 480), mode="bilinear", align_corners=True)          # lib/_utils.py:21
            loss = F.cross_entropy(logits, tgt, weight=w)
            loss.backward()
            if autocast:
                return logits.detach(), loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            return feats, lowres, logits, loss
        feats, lowres, logits, loss = run(False)
        fp32_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        named = [(k, p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()]
        dxg, dlg = x.grad.clone(), l.grad.clone()
        noise = _ref_bf16_noise(run, logits.detach(), loss.detach(), fp32_grads)
        _full_record(f"full_{tag.replace('_b4', '')}_480_b{B}", named, logits.detach(), lowres.detach(), [f.detach() for f in feats], tgt, loss.detach(),
                     {"dx": dxg, "dl": dlg}, dict(seed=1234, B=B, ws=ws, **noise))
        print(f"   ({time.time() - t0:.0f} s)")
        del model, feats, logits, loss
    import lib.video_swin_transformer as rv
    rv.sr_ratio = [1, 1, 1, 1]
    sept = ("--sep_t_pwam", "--conv3d_kernel_size_t", "3-3-3", "--conv3d_kernel_size_s", "1-1-1", "--w_t3x3_s1x1", "--mm_t3x3_s1x1")
    for tag, a in (("video_pwam", ref_args("--swin_type", "base")), ("video_sept", ref_args("--swin_type", "base", *sept))):
        if tag not in which and tag + "_fwd" not in which:
            continue
        fwd_only = tag + "_fwd" in which
        t0 = time.time()
        bb = rv.MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7),
                                            drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False,
                                            num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
        dec = rmp.SimpleDecoding(1024, a)
        model = nn.ModuleDict({"backbone": bb, "classifier": dec})
        fill_state_dict_(model)
        model.train()
        frames, l, lm, tgt = det_inputs(1, 384, 22, seed=1234, frames=8)
        if not fwd_only:
            frames.requires_grad_(True)
            l.requires_grad_(True)
        orig_in = F.instance_norm
        if tag == "video_sept":        # batch 1: sidestep PyTorch's CPU instance_norm backward defect (see _ContigGrad)
            F.instance_norm = lambda *a_, **k_: _ContigGrad.apply(orig_in(*a_, **k_))

        def run(autocast):
            model.zero_grad(set_to_none=True)
            frames.grad = l.grad = None
            with torch.set_grad_enabled(not fwd_only), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                feats = bb(frames.permute(0, 2, 1, 3, 4), l, lm)
                lowres = dec(feats[3], feats[2], feats[1], feats[0])
            with torch.set_grad_enabled(not fwd_only):
                logits = F.interpolate(lowres.float(), size=(384, 384), mode="bilinear", align_corners=True)
                loss = F.cross_entropy(logits, tgt, weight=w)
                if not fwd_only:
                    loss.backward()
            if autocast:
                return logits.detach(), loss.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            return feats, lowres, logits, loss
        try:
            feats, lowres, logits, loss = run(False)
            fp32_grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            named = [] if fwd_only else [(k, p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()]
            ing = {} if fwd_only else {"dframes": frames.grad.clone(), "dl": l.grad.clone()}
            noise = _ref_bf16_noise(run, logits.detach(), loss.detach(), fp32_grads)
        finally:
            F.instance_norm = orig_in
        _full_record(f"full_{tag}_t8_384", named, logits.detach(), lowres.detach(), [f.detach() for f in feats], tgt,
                     loss.detach(), ing, dict(seed=1234, B=1, T=8, fwd_only=int(fwd_only), **noise))
        print(f"   ({time.time() - t0:.0f} s)")
        del model, feats, logits, loss


def _ref_mask(rb, Hp, Wp, ws):
    """Run the reference's own mask construction (lib/backbone.py:634-652) by calling the
    stage forward with zero blocks and capturing the mask handed to a probe block."""
    captured = {}

    class Probe(nn.Module):
        H = W = None

        def forward(self, x, mask):
            captured["m"] = mask
            return x

    st = rb.MMBasicLayer.__new__(rb.MMBasicLayer)
    nn.Module.__init__(st)
    st.window_size, st.shift_size, st.use_checkpoint = ws, ws // 2, False
    st.blocks = nn.ModuleList([Probe()])
    st.lazy_pred, st.version, st.hs, st.downsample = False, "none", False, None
    st.fusion = lambda x, l, m: x
    st(torch.zeros(1, Hp * Wp, 1), Hp, Wp, None, None)
    return captured["m"]



def checkpoint_cases():
    """Checkpoint surgery of the reference loaders on a synthetic checkpoint (mmcv_custom/checkpoint.py:287-360, video :759-805/:830-844)."""
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    from synth_ckpt import synthetic_swin_checkpoint
    import lib.backbone as rb
    import lib.video_swin_transformer as rv
    from lib.mmcv_custom import load_checkpoint
    import logging
    a = ref_args()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        # 2-D: window-5 tables into a window-7 model, 'module.backbone.' prefixes
        path = os.path.join(td, "swin2d.pth")
        torch.save({"state_dict": synthetic_swin_checkpoint()}, path)
        bb = rb.MultiModalSwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=7, ape=False, drop_path_rate=0.0,
                                          patch_norm=True, use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
        fill_state_dict_(bb)
        load_checkpoint(bb, path, strict=False, logger=logging.getLogger("ref"))
        sd = bb.state_dict()
        for k in ("layers.0.blocks.1.attn.relative_position_bias_table", "layers.3.blocks.0.attn.relative_position_bias_table",
                  "layers.2.blocks.1.attn.qkv.weight", "patch_embed.proj.weight", "layers.1.blocks.0.mlp.fc1.weight"):
            out["swin2d|" + k] = sd[k].clone()
        # 3-D from a 2-D checkpoint (inflate) and from a 3-D checkpoint (temporal sum of the patch embedding)
        rv.sr_ratio = [1, 1, 1, 1]
        for tag, ck in (("inflate", {"model": synthetic_swin_checkpoint(prefix="")}),
                        ("video3d", {"state_dict": {("backbone." + k): v for k, v in synthetic_swin_checkpoint(prefix="", patch_t=2, ws=7, index_n=392).items()}})):
            path = os.path.join(td, tag + ".pth")
            if tag == "video3d":      # a 3-D checkpoint carries (2Wd-1)(2Wh-1)(2Ww-1) tables already
                for k in list(ck["state_dict"]):
                    if "relative_position_bias_table" in k:
                        ck["state_dict"][k] = ck["state_dict"][k].repeat(15, 1)
            torch.save(ck, path)
            b3 = rv.MultiModalSwinTransformer3D(pretrained=path, pretrained2d=(tag == "inflate"), patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2],
                                                num_heads=[1, 2, 4, 8], window_size=(8, 7, 7), drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3),
                                                use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
            fill_state_dict_(b3)
            if tag == "inflate":
                b3.inflate_weights()
            else:
                b3.init_weights()
            sd = b3.state_dict()
            for k in ("layers.0.blocks.1.attn.relative_position_bias_table", "layers.2.blocks.0.attn.qkv.weight", "patch_embed.proj.weight"):
                out[tag + "|" + k] = sd[k].clone()
    save("checkpoint_surgery", **out)

    # ---- released 2-D LAVT weights into the video model: LAVTVideo.load_from_pretrained2d_lavt_weights[_into_a_3d_model]
    # (lib/_utils.py:133-238), driven as train.py:575-578 does.  The fake 2-D checkpoint has window-5 tables (bicubic resize to 13x13, then
    # repeated 2*8-1 times) and carries relative_position_index buffers (must be dropped).
    from synth_ckpt import synthetic_lavt2d_checkpoint
    import lib._utils as ru
    from lib import mask_predictor as rmp
    m2 = ru.LAVT(rb.MultiModalSwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=5, ape=False, drop_path_rate=0.0,
                                              patch_norm=True, use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a),
                 rmp.SimpleDecoding(256, a))
    with open(os.path.join(HERE, "state_dict_keys_lavt2d_micro_w5.txt"), "w") as f:
        for k, v in m2.state_dict().items():
            f.write(f"{k}|{'x'.join(str(d) for d in v.shape)}|{'i' if not v.dtype.is_floating_point else 'f'}\n")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "lavt2d.pth")
        torch.save({"model": synthetic_lavt2d_checkpoint()}, path)
        for method in ("load_from_pretrained2d_lavt_weights", "load_from_pretrained2d_lavt_weights_into_a_3d_model"):
            b3 = rv.MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7),
                                                drop_path_rate=0.0, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False,
                                                num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=a)
            vm = ru.LAVTVideo(b3, rmp.SimpleDecoding(256, a), a)
            fill_state_dict_(vm)
            getattr(vm, method)(path)
            sd = vm.state_dict()
            for k in ("backbone.layers.0.blocks.1.attn.relative_position_bias_table", "backbone.layers.3.blocks.0.attn.relative_position_bias_table",
                      "backbone.patch_embed.proj.weight", "backbone.layers.1.blocks.0.attn.qkv.weight", "backbone.layers.1.fusion.vis_project.0.weight",
                      "backbone.layers.2.fusion.image_lang_att.f_key.0.weight", "backbone.layers.0.res_gate.0.weight", "classifier.conv2_3.weight",
                      "classifier.bn1_4.running_var"):
                out[method + "|" + k] = sd[k].clone()
    save("lavt2d_into_video", **out)


BERT_MICRO = dict(vocab_size=64, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, max_position_embeddings=32,
                  type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12, hidden_act="gelu")


def bert_inputs(B, N, vocab, seed):
    """ids / mask the way data/dataset_refer_bert.py:58-81 produces them: sentence b has a random length, zero-padded to N"""
    g = torch.Generator("cpu").manual_seed(seed)
    ids = torch.zeros(B, N, dtype=torch.long)
    mask = torch.zeros(B, N, dtype=torch.long)
    for b in range(B):
        n = N if b == 0 else int(torch.randint(3, N, (1,), generator=g))
        ids[b, :n] = torch.randint(1, vocab, (n,), generator=g)
        mask[b, :n] = 1
    return ids, mask


def bert_cases():
    """Text side: the reference's `bert.modeling_bert.BertModel` is an absent copy of HF transformers 3.0.2; the vectors come
    from the `transformers.BertModel` installed in this image, on name-keyed deterministic weights (lavt_hip.detweights)."""
    import transformers
    from transformers import BertConfig, BertModel
    m = BertModel(BertConfig(**BERT_MICRO), add_pooling_layer=False).eval()
    fill_state_dict_(m)
    for N in (20, 22):
        ids, mask = bert_inputs(2, N, BERT_MICRO["vocab_size"], seed=40 + N)
        m.zero_grad(set_to_none=True)
        out = m(ids, attention_mask=mask)[0]
        w = randn(7, *out.shape) * mask[..., None]          # the visual path only reads real-token features (PWAM masks keys / values)
        (out * w).sum().backward()
        grads = {k.replace(".", "__"): grad_digest(p.grad) for k, p in m.named_parameters()
                 if k in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight", "embeddings.LayerNorm.weight",
                          "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.key.bias",
                          "encoder.layer.1.attention.output.dense.weight", "encoder.layer.1.intermediate.dense.weight",
                          "encoder.layer.1.output.LayerNorm.bias")}
        save(f"bert_micro_n{N}", ids=ids, mask=mask, out=out, w=w, transformers_version=np.array(transformers.__version__), **{"grad__" + k: v for k, v in grads.items()})
    with open(os.path.join(HERE, "state_dict_keys_bert_micro.txt"), "w") as f:
        for k, v in m.state_dict().items():
            f.write(f"{k}|{'x'.join(str(d) for d in v.shape)}\n")


if __name__ == "__main__":
    main()
