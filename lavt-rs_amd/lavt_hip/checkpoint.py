"""Checkpoint compatibility: load public Swin / Video-Swin weights and released 2-D LAVT weights into the drop-in
modules, with the reference's key surgery.  Host-side torch code (runs once, before training); everything returns
(missing_keys, unexpected_keys) of the final non-strict `load_state_dict`.

Reference behaviour mirrored here:
* lib/mmcv_custom/checkpoint.py:287-360 `load_checkpoint`: unwrap 'state_dict' / 'model'; strip a leading 'module.'; keep only and
  strip 'backbone.' (UperNet checkpoints) or 'encoder.' (MoBY); bicubic-resize `relative_position_bias_table` when the window differs;
  non-strict load.
* lib/video_swin_transformer.py:759-805 `inflate_weights` (2-D Swin -> 3-D): drop `relative_position_index` / `attn_mask`; patch-embed
  weight `unsqueeze(2).repeat(patch_t) / patch_t`; bias tables resized to (2Wh-1, 2Ww-1) then repeated (2Wd-1) times along the table axis.
* lib/video_swin_transformer.py:830-844 `init_weights` (3-D checkpoint): take 'state_dict', keep keys containing 'backbone.' with the
  first 9 characters removed, SUM the patch-embed weight over its temporal axis (keepdim).
* lib/_utils.py:133-238 (2-D LAVT weights into the video model): patch-embed `unsqueeze(2)`, tables as in inflate_weights, optionally
  dropping the '.fusion' tensors.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


def _unwrap(checkpoint):
    if not isinstance(checkpoint, dict):
        raise RuntimeError("no state_dict found in checkpoint")
    if "state_dict" in checkpoint:
        return checkpoint["state_dict"]
    if "model" in checkpoint:
        return checkpoint["model"]
    return checkpoint


def _resize_table(table, size_hw):
    """(L1, nH) -> (size_h*size_w, nH), bicubic over the square (S1, S1) source grid (mmcv_custom/checkpoint.py:347-353)"""
    L1, nH = table.shape
    S1 = int(L1 ** 0.5)
    r = F.interpolate(table.permute(1, 0).view(1, nH, S1, S1), size=tuple(size_hw), mode="bicubic")
    return r.view(nH, size_hw[0] * size_hw[1]).permute(1, 0)


def convert_swin_state_dict(checkpoint, model):
    """2-D Swin checkpoint -> state dict for lib.backbone.MultiModalSwinTransformer (reference load_checkpoint semantics)."""
    sd = _unwrap(checkpoint)
    keys = list(sd.keys())
    if keys[0].startswith("module."):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    if list(sd.keys())[0].startswith("backbone."):
        sd = OrderedDict((k.replace("backbone.", ""), v) for k, v in sd.items() if k.startswith("backbone."))
    if sorted(sd.keys())[0].startswith("encoder"):
        sd = OrderedDict((k.replace("encoder.", ""), v) for k, v in sd.items() if k.startswith("encoder."))
    sd = OrderedDict(sd)
    sd.pop("absolute_pos_embed", None)                       # ape=False in every LAVT factory
    current = model.state_dict()
    for k in [k for k in sd if "relative_position_bias_table" in k]:
        if k not in current:
            continue
        L1, nH1 = sd[k].shape
        L2, nH2 = current[k].shape
        if nH1 != nH2:
            print(f"Error in loading {k}, pass")
        elif L1 != L2:
            S2 = int(L2 ** 0.5)
            sd[k] = _resize_table(sd[k], (S2, S2))
    return sd


def _load_tolerant(model, sd):
    """mmcv's load_state_dict (mmcv_custom/checkpoint.py:41-108) reports tensors of the wrong shape instead of raising; the constant
    `relative_position_index` / `attn_mask` buffers of a checkpoint are never needed."""
    current = model.state_dict()
    keep = OrderedDict()
    for k, v in sd.items():
        if "relative_position_index" in k or "attn_mask" in k:
            continue
        if k in current and tuple(current[k].shape) != tuple(v.shape):
            print(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(current[k].shape)}, skipped")
            continue
        keep[k] = v
    res = model.load_state_dict(keep, strict=False)
    return [k for k in res.missing_keys if "relative_position_index" not in k], list(res.unexpected_keys)


def load_swin_checkpoint(model, filename, map_location="cpu"):
    return _load_tolerant(model, convert_swin_state_dict(torch.load(filename, map_location=map_location, weights_only=False), model))


def _tables_to_3d(sd, model, window_size):
    wd, wh, ww = window_size
    current = model.state_dict()
    for k in [k for k in sd if "relative_position_bias_table" in k]:
        t = sd[k]
        L1, nH1 = t.shape
        nH2 = current[k].shape[1] if k in current else nH1
        L2 = (2 * wh - 1) * (2 * ww - 1)
        if nH1 != nH2:
            print(f"Error in loading {k}, passing")
        elif L1 != L2:
            t = _resize_table(t, (2 * wh - 1, 2 * ww - 1))
        sd[k] = t.repeat(2 * wd - 1, 1)
    return sd


def convert_video_swin_state_dict(checkpoint, model, inflate_2d=False):
    """-> state dict for lib.video_swin_transformer.MultiModalSwinTransformer3D"""
    if inflate_2d:
        sd = OrderedDict(checkpoint["model"])
        for k in [k for k in sd if "relative_position_index" in k or "attn_mask" in k]:
            del sd[k]
        pt = model.patch_size[0]
        sd["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].unsqueeze(2).repeat(1, 1, pt, 1, 1) / pt
        return _tables_to_3d(sd, model, model.window_size)
    sd = OrderedDict((k[9:], v) for k, v in checkpoint["state_dict"].items() if "backbone." in k)
    sd["patch_embed.proj.weight"] = sd["patch_embed.proj.weight"].sum(dim=2, keepdim=True)
    return sd


def load_video_swin_checkpoint(model, filename, inflate_2d=False, map_location="cpu"):
    return _load_tolerant(model, convert_video_swin_state_dict(torch.load(filename, map_location=map_location, weights_only=False), model, inflate_2d))


def convert_lavt2d_to_video_state_dict(checkpoint, video_model, drop_fusion=False):
    """released 2-D LAVT weights ('model' key, 'backbone.' / 'classifier.' prefixes) -> LAVTVideo (lib/_utils.py:133-238)"""
    sd = OrderedDict(checkpoint["model"])
    for k in [k for k in sd if "relative_position_index" in k or "attn_mask" in k or (drop_fusion and ".fusion" in k)]:
        del sd[k]
    assert "backbone.patch_embed.proj.weight" in sd
    sd["backbone.patch_embed.proj.weight"] = sd["backbone.patch_embed.proj.weight"].unsqueeze(2)
    return _tables_to_3d(sd, video_model, video_model.backbone.window_size)


def load_lavt2d_into_video(video_model, filename, drop_fusion=False, map_location="cpu"):
    return _load_tolerant(video_model, convert_lavt2d_to_video_state_dict(torch.load(filename, map_location=map_location, weights_only=False), video_model, drop_fusion))
