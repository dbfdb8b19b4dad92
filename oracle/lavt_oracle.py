"""CPU oracle for the LAVT hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference's algorithm for the
path `Swin backbone + PWAM + language gate + SimpleDecoding + bilinear upsample`, written
from the maths in SURVEY.md Appendix A (not from the reference's code).  It exists so that
tests can check the HIP path on a machine where /root/reference does not exist.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product (lavt-rs_amd/) never does; the product fails loudly without its HIP library.

Parity pin: tests/test_oracle_golden.py checks every function here against fixtures in
tests/golden/ that were produced by importing the real reference
(tests/golden/make_golden.py, run in the build container).

Functional style: every function takes `sd` (a state dict with the reference's key names,
SURVEY.md 8b) and a key prefix.  Autograd through these functions is the gradient oracle.

Reference anchors (file:line in /root/reference):
  patch_embed            lib/backbone.py:315-331
  swin_block             lib/backbone.py:188-245
  window_attention       lib/backbone.py:113-143 (+ index :89-103)
  shift_mask             lib/backbone.py:634-652
  patch_merging          lib/backbone.py:261-288
  pwam / sila            lib/backbone.py:1265-1278 / :1329-1372
  stage (gate, outputs)  lib/backbone.py:625-686
  backbone               lib/backbone.py:490-515
  decoder                lib/mask_predictor.py:56-99
  lavt_forward           lib/_utils.py:16-23
  weighted_ce            losses.py:7-11
  iou_counts             train.py:64-76, test.py:242-246
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5

VARIANTS = {  # lib/segmentation.py:85-100
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
    "small": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    "large": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48)),
    "micro": dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8)),   # test-only size (not in the reference)
}


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def _lin(x, sd, p, bias=True):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _pw(x, sd, p):
    """1x1 Conv1d on token-major data: x (B,T,Cin), weight (Cout,Cin,1)."""
    return F.linear(x, sd[p + ".weight"][:, :, 0], sd[p + ".bias"])


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# ----------------------------------------------------------------------------- A1
def patch_embed(sd, p, img):
    """img (B,3,H,W) -> tokens (B, H4*W4, C0), H4, W4.  Zero-pad to x4, 4x4/4 conv, LN."""
    B, _, H, W = img.shape
    img = F.pad(img, (0, (-W) % 4, 0, (-H) % 4))
    y = F.conv2d(img, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=4)
    H4, W4 = y.shape[2:]
    t = y.flatten(2).transpose(1, 2)
    if (p + ".norm.weight") in sd:
        t = _ln(t, sd, p + ".norm")
    return t, H4, W4


# ----------------------------------------------------------------------------- A3/A4
def rel_pos_index(ws):
    r = torch.arange(ws)
    rr, cc = torch.meshgrid(r, r, indexing="ij")
    rr, cc = rr.reshape(-1), cc.reshape(-1)
    return (rr[:, None] - rr[None, :] + ws - 1) * (2 * ws - 1) + (cc[:, None] - cc[None, :] + ws - 1)


def shift_mask(Hp, Wp, ws, s):
    """(nW, N, N) additive mask, 0 / -100.0, regions on the padded grid (A4)."""
    def g(n, size):
        v = torch.full((size,), 2, dtype=torch.long)
        v[: size - s] = 1
        v[: size - ws] = 0
        return v
    ids = 3 * g(0, Hp)[:, None] + g(0, Wp)[None, :]
    idw = ids.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return torch.where(idw[:, :, None] == idw[:, None, :], 0.0, -100.0)


def window_attention(sd, p, xw, nH, ws, mask=None):
    """xw (B_, N, C) windows -> (B_, N, C)."""
    B_, N, C = xw.shape
    hd = C // nH
    qkv = _lin(xw, sd, p + ".qkv").view(B_, N, 3, nH, hd)
    q = qkv[:, :, 0].transpose(1, 2) * hd ** -0.5
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    a = q @ k.transpose(-1, -2)                                            # (B_,nH,N,N)
    bias = sd[p + ".relative_position_bias_table"][rel_pos_index(ws).reshape(-1)]
    a = a + bias.view(N, N, nH).permute(2, 0, 1)[None]
    if mask is not None:
        nW = mask.shape[0]
        a = (a.view(B_ // nW, nW, nH, N, N) + mask[None, :, None]).view(B_, nH, N, N)
    a = torch.softmax(a, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B_, N, C)
    return _lin(o, sd, p + ".proj")


def swin_block(sd, p, x, H, W, nH, ws, shifted):
    """x (B, H*W, C).  A2."""
    B, T, C = x.shape
    s = ws // 2 if shifted else 0
    u = _ln(x, sd, p + ".norm1").view(B, H, W, C)
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    u = F.pad(u, (0, 0, 0, Wp - W, 0, Hp - H))
    if s:
        u = torch.roll(u, (-s, -s), (1, 2))
    win = u.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    mask = shift_mask(Hp, Wp, ws, s) if s else None
    a = window_attention(sd, p + ".attn", win, nH, ws, mask)
    a = a.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if s:
        a = torch.roll(a, (s, s), (1, 2))
    x = x + a[:, :H, :W].reshape(B, T, C)
    h = gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1"))
    return x + _lin(h, sd, p + ".mlp.fc2")


# ----------------------------------------------------------------------------- A5
def patch_merging(sd, p, x, H, W):
    B, T, C = x.shape
    x = x.view(B, H, W, C)
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    z = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    z = z.reshape(B, -1, 4 * C)
    return _lin(_ln(z, sd, p + ".norm"), sd, p + ".reduction", bias=False)


# ----------------------------------------------------------------------------- A6
def instance_norm_tokens(z):
    """Per (batch, channel) normalisation over all positions; z (B, T, C); biased var, eps 1e-5."""
    mu = z.mean(1, keepdim=True)
    var = z.var(1, unbiased=False, keepdim=True)
    return (z - mu) / torch.sqrt(var + 1e-5)


def sila(sd, p, x, l, m, G=1):
    """SpatialImageLanguageAttention.  x (B,T,C), l (B,768,Nl), m (B,Nl,1) -> (B,T,C)."""
    B, T, C = x.shape
    lt = l.transpose(1, 2)                                                 # (B,Nl,768)
    q = instance_norm_tokens(_pw(x, sd, p + ".f_query.0"))                 # (B,T,C)
    k = _pw(lt, sd, p + ".f_key.0") * m                                    # (B,Nl,C)
    v = _pw(lt, sd, p + ".f_value.0") * m
    Nl = k.shape[1]
    qh = q.view(B, T, G, C // G).transpose(1, 2)                           # (B,G,T,c)
    kh = k.view(B, Nl, G, C // G).permute(0, 2, 3, 1)                      # (B,G,c,Nl)
    vh = v.view(B, Nl, G, C // G).transpose(1, 2)                          # (B,G,Nl,c)
    s = (qh @ kh) * C ** -0.5 + (1e4 * m.transpose(1, 2)[:, None] - 1e4)   # scale uses full C
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, C)
    return instance_norm_tokens(_pw(o, sd, p + ".W.0"))


def pwam(sd, p, x, l, m, G=1):
    vis = gelu(_pw(x, sd, p + ".vis_project.0"))
    lang = sila(sd, p + ".image_lang_att", x, l, m, G)
    return gelu(_pw(vis * lang, sd, p + ".project_mm.0"))


# ----------------------------------------------------------------------------- A7
def stage(sd, p, x, H, W, l, m, depth, nH, ws, last, G=1):
    """Returns (stage feature r, x for the next stage, next H, next W)."""
    for b in range(depth):
        x = swin_block(sd, f"{p}.blocks.{b}", x, H, W, nH, ws, shifted=(b % 2 == 1))
    r = pwam(sd, p + ".fusion", x, l, m, G)
    g = torch.tanh(F.linear(F.relu(F.linear(r, sd[p + ".res_gate.0.weight"])), sd[p + ".res_gate.2.weight"]))
    x = x + g * r
    if last:
        return r, x, H, W
    return r, patch_merging(sd, p + ".downsample", x, H, W), (H + 1) // 2, (W + 1) // 2


def backbone(sd, p, img, l, m, variant="tiny", ws=7, mha=(1, 1, 1, 1)):
    """-> 4 NCHW feature maps (c1..c4)."""
    cfg = VARIANTS[variant]
    x, H, W = patch_embed(sd, p + ".patch_embed", img)
    outs = []
    for i in range(4):
        r, x, Hn, Wn = stage(sd, f"{p}.layers.{i}", x, H, W, l, m, cfg["depths"][i], cfg["num_heads"][i], ws,
                             last=(i == 3), G=mha[i])
        f = _ln(r, sd, f"{p}.norm{i}")
        outs.append(f.view(-1, H, W, f.shape[-1]).permute(0, 3, 1, 2).contiguous())
        H, W = Hn, Wn
    return tuple(outs)


# ----------------------------------------------------------------------------- A8
def _bn(x, sd, p, training):
    if training:   # batch statistics, biased variance; running stats not updated by the oracle
        mu = x.mean((0, 2, 3), keepdim=True)
        var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    else:
        mu = sd[p + ".running_mean"].view(1, -1, 1, 1)
        var = sd[p + ".running_var"].view(1, -1, 1, 1)
    return (x - mu) / torch.sqrt(var + 1e-5) * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)


def decoder(sd, p, c4, c3, c2, c1, training=False):
    x = c4
    for lvl, skip in ((4, c3), (3, c2), (2, c1)):
        if x.shape[-2] < skip.shape[-2] or x.shape[-1] < skip.shape[-1]:
            x = F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=True)
        x = torch.cat([x, skip], 1)
        x = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1_{lvl}.weight"], padding=1), sd, f"{p}.bn1_{lvl}", training))
        x = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv2_{lvl}.weight"], padding=1), sd, f"{p}.bn2_{lvl}", training))
    return F.conv2d(x, sd[p + ".conv1_1.weight"], sd[p + ".conv1_1.bias"])


def lavt_forward(sd, img, l, m, variant="tiny", ws=7, training=False, mha=(1, 1, 1, 1)):
    """LAVT.forward: logits (B,2,H,W)."""
    c1, c2, c3, c4 = backbone(sd, "backbone", img, l, m, variant, ws, mha)
    y = decoder(sd, "classifier", c4, c3, c2, c1, training)
    return F.interpolate(y, size=img.shape[-2:], mode="bilinear", align_corners=True)


# ----------------------------------------------------------------------------- A10
def weighted_ce(logits, target):
    return F.cross_entropy(logits, target, weight=torch.tensor([0.9, 1.1], dtype=logits.dtype))


def iou_counts(logits, target):
    pred = logits.argmax(1)
    inter = int((pred & target).sum())
    union = int((pred | target).sum())
    return inter, union


def multiclass_dice(logits, target, eps=1e-6):
    """MultiClassDiceLoss.forward (reference losses.py:38-77): softmax over classes; per sample and class
    I = sum p*onehot, C = sum (p*p + onehot) over pixels; mean over samples of (1 - 2I/(C+eps)), then the mean of the two classes."""
    p = torch.softmax(logits, 1)
    oh = F.one_hot(target, logits.shape[1]).permute(0, 3, 1, 2).to(logits.dtype)
    inter = (p * oh).sum((2, 3))
    card = (p * p + oh).sum((2, 3))
    dl = (1.0 - 2.0 * inter / (card + eps)).mean(0)
    return (dl[1] + dl[0]) / 2
