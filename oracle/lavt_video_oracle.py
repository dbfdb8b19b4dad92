"""CPU oracle for the VIDEO path of LAVT (Video-Swin backbone + PWAM / SepTPWAM + gate + the shared 2-D decoder)
-- TEST INFRASTRUCTURE ONLY (same rules as oracle/lavt_oracle.py: never imported by the product).

Plain-PyTorch fp32 restatement written from SURVEY.md Appendix A9 / section 8 a16-a17; pinned to golden vectors captured from
the real reference by tests/golden/make_golden.py (tests/test_oracle_golden.py::test_video_*).

Reference anchors (lib/video_swin_transformer.py): get_window_size :70-83, compute_mask :315-328, window_partition/reverse :39-67,
WindowAttention3D :137-168 (index slice quirk :150), SwinTransformerBlock3D :214-273, PatchMerging :289-311, PatchEmbed3D :616-634,
MMBasicLayer.forward :538-591, MultiModalSwinTransformer3D.forward :854-881, PWAM :889-1009, SepTPWAM :1480-1584;
lib/_utils.py:86-108 (LAVTVideo.forward).
"""
import torch
import torch.nn.functional as F

from . import lavt_oracle as O2

VIDEO_VARIANTS = {   # lib/segmentation.py:156-172 (embed_dim, depths, heads, drop_path_rate)
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
    "small": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    "micro": dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8)),      # test-only
}


def clip_window(size, window, shift=None):
    """per axis: a window larger than (or equal to) the feature is clipped to it and its shift zeroed"""
    w = [min(s, ws) if s <= ws else ws for s, ws in zip(size, window)]
    if shift is None:
        return tuple(w)
    sh = [0 if s <= ws else shv for s, ws, shv in zip(size, window, shift)]
    return tuple(w), tuple(sh)


def rel_pos_index_3d(wd, wh, ww):
    d, h, w = torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")
    d, h, w = d.reshape(-1), h.reshape(-1), w.reshape(-1)
    return ((d[:, None] - d[None] + wd - 1) * (2 * wh - 1) + (h[:, None] - h[None] + wh - 1)) * (2 * ww - 1) + (w[:, None] - w[None] + ww - 1)


def shift_mask_3d(Dp, Hp, Wp, win, shift):
    """(nW, N, N) 0 / -100.  Axis label: 0 | 1 | 2 for [0,n-w) | [n-w,n-s) | [n-s,n); with s == 0 every position is in the last slice."""
    def g(n, w, s):
        v = torch.full((n,), 2, dtype=torch.long)
        if s > 0:
            v[: n - s] = 1
            v[: n - w] = 0
        return v
    ids = 9 * g(Dp, win[0], shift[0])[:, None, None] + 3 * g(Hp, win[1], shift[1])[None, :, None] + g(Wp, win[2], shift[2])[None, None, :]
    idw = ids.view(Dp // win[0], win[0], Hp // win[1], win[1], Wp // win[2], win[2]).permute(0, 2, 4, 1, 3, 5).reshape(-1, win[0] * win[1] * win[2])
    return torch.where(idw[:, :, None] == idw[:, None, :], 0.0, -100.0)


def window_attention_3d(sd, p, xw, nH, full_window, mask=None):
    """xw (B_, N, C); the bias index is the top-left N x N block of the FULL window's index matrix (reference quirk)."""
    B_, N, C = xw.shape
    hd = C // nH
    qkv = F.linear(xw, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias")).view(B_, N, 3, nH, hd)
    q = qkv[:, :, 0].transpose(1, 2) * hd ** -0.5
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    a = q @ k.transpose(-1, -2)
    idx = rel_pos_index_3d(*full_window)[:N, :N].reshape(-1)
    a = a + sd[p + ".relative_position_bias_table"][idx].view(N, N, nH).permute(2, 0, 1)[None]
    if mask is not None:
        nW = mask.shape[0]
        a = (a.view(B_ // nW, nW, nH, N, N) + mask[None, :, None]).view(B_, nH, N, N)
    o = (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def swin_block_3d(sd, p, x, nH, window, shifted):
    """x (B, D, H, W, C)"""
    B, D, H, W, C = x.shape
    shift_full = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
    win, shift = clip_window((D, H, W), window, shift_full)
    u = O2._ln(x, sd, p + ".norm1")
    Dp, Hp, Wp = (-(-n // w) * w for n, w in zip((D, H, W), win))
    u = F.pad(u, (0, 0, 0, Wp - W, 0, Hp - H, 0, Dp - D))
    moved = any(s > 0 for s in shift)
    if moved:
        u = torch.roll(u, tuple(-s for s in shift), (1, 2, 3))
    xw = u.view(B, Dp // win[0], win[0], Hp // win[1], win[1], Wp // win[2], win[2], C).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, win[0] * win[1] * win[2], C)
    mask = shift_mask_3d(Dp, Hp, Wp, win, shift) if moved else None
    a = window_attention_3d(sd, p + ".attn", xw, nH, window, mask)
    a = a.view(B, Dp // win[0], Hp // win[1], Wp // win[2], win[0], win[1], win[2], C).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, Dp, Hp, Wp, C)
    if moved:
        a = torch.roll(a, shift, (1, 2, 3))
    x = x + a[:, :D, :H, :W]
    h = O2.gelu(O2._lin(O2._ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1"))
    return x + O2._lin(h, sd, p + ".mlp.fc2")


def patch_merging_3d(sd, p, x):
    """(B, D, H, W, C) -> (B, D, ceil(H/2), ceil(W/2), 2C): spatial only"""
    B, D, H, W, C = x.shape
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    z = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    return O2._lin(O2._ln(z, sd, p + ".norm"), sd, p + ".reduction", bias=False)


def _in3(z):
    """InstanceNorm3d on channels-last (B, D, H, W, C): per (b, c) over all D*H*W"""
    mu = z.mean((1, 2, 3), keepdim=True)
    var = z.var((1, 2, 3), unbiased=False, keepdim=True)
    return (z - mu) / torch.sqrt(var + 1e-5)


def _conv3(x, sd, p):
    """Conv3d with 'same' padding on channels-last (B, D, H, W, C)"""
    w = sd[p + ".weight"]
    pad = tuple(k // 2 for k in w.shape[2:])
    return F.conv3d(x.permute(0, 4, 1, 2, 3), w, sd[p + ".bias"], padding=pad).permute(0, 2, 3, 4, 1)


def sep_t_pwam(sd, p, x, l, m, G=1):
    """README video recipe (w_t3x3_s1x1 + mm_t3x3_s1x1).  x (B, D, H, W, C) -> (B, D*H*W, C)."""
    B, D, H, W, C = x.shape
    vis = O2.gelu(_conv3(x, sd, p + ".temporal_vis_project.0")) + O2.gelu(_conv3(x, sd, p + ".spatial_vis_project.0"))
    q = (_in3(_conv3(x, sd, p + ".f_query_t.0")) + _in3(_conv3(x, sd, p + ".f_query_s.0"))).reshape(B, D * H * W, C)
    lt = l.transpose(1, 2)
    k = O2._pw(lt, sd, p + ".f_key.0") * m
    v = O2._pw(lt, sd, p + ".f_value.0") * m
    Nl = k.shape[1]
    qh = q.view(B, -1, G, C // G).transpose(1, 2)
    kh = k.view(B, Nl, G, C // G).permute(0, 2, 3, 1)
    vh = v.view(B, Nl, G, C // G).transpose(1, 2)
    s = (qh @ kh) * C ** -0.5 + (1e4 * m.transpose(1, 2)[:, None] - 1e4)
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, D, H, W, C)
    lang = _in3(_conv3(o, sd, p + ".W_t.0")) + _in3(_conv3(o, sd, p + ".W_s.0"))
    mm = vis * lang
    out = O2.gelu(_conv3(mm, sd, p + ".project_mm_t.0")) + O2.gelu(_conv3(mm, sd, p + ".project_mm_s.0"))
    return out.reshape(B, D * H * W, C)


def stage_3d(sd, p, x, l, m, depth, nH, window, last, sep_t=False, G=1):
    """x (B, D, H, W, C) -> (stage feature r (B,D,H,W,C), x for the next stage)"""
    B, D, H, W, C = x.shape
    for b in range(depth):
        x = swin_block_3d(sd, f"{p}.blocks.{b}", x, nH, window, shifted=(b % 2 == 1))
    if sep_t:
        r = sep_t_pwam(sd, p + ".fusion", x, l, m, G)
    else:
        r = O2.pwam(sd, p + ".fusion", x.reshape(B, D * H * W, C), l, m, G)
    xt = x.reshape(B, D * H * W, C)
    g = torch.tanh(F.linear(F.relu(F.linear(r, sd[p + ".res_gate.0.weight"])), sd[p + ".res_gate.2.weight"]))
    xt = (xt + g * r).view(B, D, H, W, C)
    r = r.view(B, D, H, W, C)
    return r, (xt if last else patch_merging_3d(sd, p + ".downsample", xt))


def backbone_3d(sd, p, vid, l, m, variant="micro", window=(8, 7, 7), sep_t=False):
    """vid (B, 3, T, H, W) -> 4 maps (B*T, C_i, H_i, W_i)"""
    cfg = VIDEO_VARIANTS[variant]
    B, _, T, H, W = vid.shape
    vid = F.pad(vid, (0, (-W) % 4, 0, (-H) % 4))
    y = F.conv3d(vid, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=(1, 4, 4))   # (B, C0, T, H4, W4)
    x = O2._ln(y.permute(0, 2, 3, 4, 1), sd, p + ".patch_embed.norm")
    outs = []
    for i in range(4):
        r, x = stage_3d(sd, f"{p}.layers.{i}", x, l, m, cfg["depths"][i], cfg["num_heads"][i], window, last=(i == 3), sep_t=sep_t)
        f = O2._ln(r, sd, f"{p}.norm{i}")
        outs.append(f.reshape(-1, f.shape[2], f.shape[3], f.shape[4]).permute(0, 3, 1, 2).contiguous())
    return tuple(outs)


def lavt_video_forward(sd, frames, l, m, variant="micro", window=(8, 7, 7), sep_t=False, training=False):
    """frames (B, T, 3, H, W); l (B, 768, Nl) language features (the BERT output); m (B, Nl, 1) -> logits (B*T, 2, H, W)"""
    c1, c2, c3, c4 = backbone_3d(sd, "backbone", frames.permute(0, 2, 1, 3, 4), l, m, variant, window, sep_t)
    y = O2.decoder(sd, "classifier", c4, c3, c2, c1, training)
    return F.interpolate(y, size=frames.shape[-2:], mode="bilinear", align_corners=True)
