"""MI355X-native multi-modal Swin backbone of LAVT (2-D image path).

Drop-in for the reference's lib/backbone.py on the hot path: same class names, constructor
arguments, sub-module names (=> identical state-dict keys) and forward signatures
(`MultiModalSwinTransformer.forward(x, l, l_mask) -> 4 NCHW-shaped feature maps`,
reference lib/backbone.py:490-515), but nothing here calls a torch.nn forward: the standard
nn.Linear / nn.Conv1d / nn.LayerNorm objects are parameter containers only and every
computation is a liblavt_hip kernel launched through lavt_hip.ops.

Data layout: tokens stay [B*H*W, C] (NHWC) from the patch embedding to the stage outputs; the
pad/roll/window_partition/window_reverse/roll/crop sequence of the reference
(lib/backbone.py:204-237) does not exist as data movement -- it is a row map consumed by the qkv
GEMM's loads and the proj GEMM's stores.  Out of scope here: BCAM/GACD/EFN fusion,
LangProject (`--fuse simple`), the 2-D-Swin-on-video ablation layers, the language-free Swin.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from lavt_hip import ops, rowmaps
from lavt_hip._capi import ACT_GELU, ACT_NONE, ACT_RELU, scope
from lavt_hip.runtime import compute_dtype


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class _LangCtx:
    """Per-forward language side tensors shared by the four PWAMs (built once per (l, l_mask) pair)."""
    _cache = None

    def __init__(self, l, l_mask, dtype):
        B, Cl, n_l = l.shape
        if n_l > ops.KV_LD:
            raise ValueError(f"PWAM supports at most {ops.KV_LD} language tokens, got {n_l}")
        self.B, self.n_l = B, n_l
        self.lt = ops.transpose_last2(l.float(), dtype).view(B * n_l, Cl)            # (B*N_l, 768) token-major
        self.mask_rows, self.maskbias = ops.lang_mask(l_mask, B, n_l)                 # float mask rows, 1e4 * m - 1e4 (lib/backbone.py:1360): one launch
        self.kv_map = rowmaps.kv_pad_map(B, n_l, ops.KV_LD, l.device)
        self.plan, self.kv_done = None, None

    def set_plan(self, layers):
        """layers: the (f_key, f_value) 1x1 convolutions of every stage's fusion module, in stage order.  With a plan the key / value projections
        of all stages are computed by ONE GEMM at the first request of a forward (ops.kv_all); without one (a fusion module called on its own)
        each module projects its own."""
        self.plan, self.kv_done = (list(layers) if layers is not None else None), None

    def kv(self, f_key, f_value):
        """-> (k, v, gradient sinks or None) for this module's projections"""
        if self.plan is not None and os.environ.get("LAVT_KV_HOIST", "1") != "0":
            if self.kv_done is None:
                self.kv_done = ops.kv_all(self.lt, self, self.plan)
            for (fk, fv), res in zip(self.plan, self.kv_done):
                if fk is f_key and fv is f_value:
                    return res
        kw = dict(out_map=self.kv_map, out_rows=self.B * ops.KV_LD, zero_init=True, row_scale=self.mask_rows, row_scale_value=1.0)
        return (ops.linear(self.lt, f_key.weight, f_key.bias, **kw), ops.linear(self.lt, f_value.weight, f_value.bias, **kw), None)

    @classmethod
    def get(cls, l, l_mask, dtype):
        key = (id(l), id(l_mask), l._version, l_mask._version, dtype)
        if cls._cache is None or cls._cache[0] != key or cls._cache[2] is not l or cls._cache[3] is not l_mask:
            cls._cache = (key, cls(l, l_mask, dtype), l, l_mask)
        return cls._cache[1]


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU or drop != 0.:
            raise NotImplementedError("liblavt_hip Mlp: GELU without dropout only (reference defaults)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, residual=None, row_scale=None, row_scale_div=1, row_scale_value=0.0):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y = ops.mlp(x2, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual=residual, row_scale=row_scale,
                    row_scale_div=row_scale_div, row_scale_value=row_scale_value)
        return y.view(*shp[:-1], y.shape[-1])


class WindowAttention(nn.Module):
    """W-MSA parameters + core.  Reference: lib/backbone.py:65-143."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, _pair(window_size), num_heads
        if dim // num_heads != 32 or qk_scale is not None or attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("liblavt_hip WindowAttention: head_dim 32, default scale, no dropout")
        ws = self.window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, num_heads))
        r = np.arange(ws)
        rr, cc = (a.reshape(-1) for a in np.meshgrid(r, r, indexing="ij"))
        idx = (rr[:, None] - rr[None, :] + ws - 1) * (2 * ws - 1) + (cc[:, None] - cc[None, :] + ws - 1)
        self.register_buffer("relative_position_index", torch.from_numpy(idx.astype(np.int64)))   # state-dict compatibility only
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, mask=None):
        """x: (num_windows*B, N, C) already partitioned windows.  `mask`: None, or the int8 region table
        (nW, N) of lavt_hip.rowmaps.region_ids (the dense 0/-100 float mask of the reference is never built)."""
        if mask is not None and mask.dtype != torch.int8:
            raise TypeError("pass the int8 region table (rowmaps.region_ids), not a dense float mask")
        B_, N, C = x.shape
        ws = self.window_size[0]
        qkv = ops.linear(x.reshape(B_ * N, C), self.qkv.weight, self.qkv.bias)
        a = ops.window_attention(qkv, self.relative_position_bias_table, mask, ws, self.num_heads)
        return ops.linear(a, self.proj.weight, self.proj.bias).view(B_, N, C)


class DropPath(nn.Module):
    """Per-sample stochastic depth factor (timm semantics: floor(keep + U[0,1)) / keep); identity in eval."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def factors(self, B, device):
        if self.drop_prob == 0. or not self.training:
            return None
        pre = getattr(self, "_batched", None)          # the backbone draws all blocks' factors in one shot per forward
        if pre:
            return pre.pop(0)
        keep = 1.0 - self.drop_prob
        return torch.floor(keep + torch.rand(B, device=device, dtype=torch.float32)) / keep


class SwinTransformerBlock(nn.Module):
    """Reference: lib/backbone.py:146-245.  x: (B, H*W, C) with self.H / self.W set by the stage."""

    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size, self.mlp_ratio = dim, num_heads, window_size, shift_size, mlp_ratio
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, _pair(window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None

    def forward(self, x, mask_matrix=None):
        B, L, C = x.shape
        H, W, ws, s = self.H, self.W, self.window_size, self.shift_size
        assert L == H * W, "input feature has wrong size"
        dev = x.device
        x2 = x.reshape(B * L, C)
        wmap = rowmaps.window_map(B, H, W, ws, s, dev)
        region = rowmaps.region_ids(H, W, ws, s, dev) if s > 0 else None
        M = wmap.numel()
        a = self.attn
        f1 = self.drop_path.factors(B, dev)
        dpv = 1.0 / (1.0 - self.drop_path.drop_prob) if self.drop_path.drop_prob < 1.0 else 0.0
        with scope("wmsa"):
            padded_win = M > B * L                       # padded window positions exist: weight gradients contract over the tokens instead of the window rows
            winv = rowmaps.window_inverse(B, H, W, ws, s, dev) if padded_win else None
            if ops.wmsa_fused_ok(x2, ws, self.num_heads, a.qkv.bias is not None):
                # norm1 + partition + qkv + attention as one kernel (csrc/wmsa_fused.hip): the LayerNorm and qkv GEMM launches are gone from the forward
                o, x2 = ops.wmsa_fused(x2, self.norm1, a, region, wmap, ws, self.num_heads,
                                       tok=(winv, rowmaps.window_pad_rows(B, H, W, ws, s, dev)) if padded_win else None)
            else:
                xn, x2 = ops.layer_norm_res(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps)     # x2: alias for the residual branch
                qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias, in_map=wmap, rows=M)
                o = ops.window_attention(qkv, a.relative_position_bias_table, region, ws, self.num_heads)
            x2 = ops.linear(o, a.proj.weight, a.proj.bias, residual=x2, out_map=wmap, rows=M, out_rows=B * L,
                            row_scale=f1, row_scale_div=M // B, row_scale_value=dpv, out_inv=winv)
        f2 = self.drop_path.factors(B, dev)
        m = self.mlp
        with scope("mlp"):
            if ops.ln_mlp_ok(x2, m.fc1.weight, m.fc1.bias, m.fc2.weight):
                # norm2 folded into fc1's contraction (lavt_gemm_nt.ln_wsum): no LayerNorm launch, no [M, C] LayerNorm output in the forward
                x2 = ops.ln_mlp(x2, self.norm2, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, row_scale=f2, row_scale_div=L, row_scale_value=dpv)
            else:
                h, x2 = ops.layer_norm_res(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
                x2 = self.mlp(h, residual=x2, row_scale=f2, row_scale_div=L, row_scale_value=dpv)
        return x2.view(B, L, C)


class PatchMerging(nn.Module):
    """Reference: lib/backbone.py:248-288.  The 2x2 gather is fused into the LayerNorm kernel."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        g = rowmaps.merge_map(B, H, W, x.device)
        z = ops.layer_norm(x.reshape(B * L, C), self.norm.weight, self.norm.bias, self.norm.eps, gather=g)
        y = ops.linear(z, self.reduction.weight, None)
        return y.view(B, -1, 2 * C)


class PatchEmbed(nn.Module):
    """Reference: lib/backbone.py:291-331.  Returns (B, C, Wh, Ww)-shaped tensor (NHWC memory)."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        if _pair(patch_size) != (4, 4) or in_chans != 3:
            raise NotImplementedError("liblavt_hip PatchEmbed: 4x4 patches of 3-channel images")
        self.patch_size, self.in_chans, self.embed_dim = _pair(patch_size), in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=4, stride=4)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def tokens(self, x, dtype):
        B, _, H, W = x.shape
        t = ops.patch_embed(x, self.proj.weight, self.proj.bias, dtype)
        if self.norm is not None:
            t = ops.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps)
        return t, (H + 3) // 4, (W + 3) // 4

    def forward(self, x):
        t, Wh, Ww = self.tokens(x, compute_dtype())
        return t.view(x.shape[0], Wh, Ww, self.embed_dim).permute(0, 3, 1, 2)


class SpatialImageLanguageAttention(nn.Module):
    """Reference: lib/backbone.py:1281-1372 (InstanceNorm variant only)."""

    def __init__(self, v_in_channels, l_in_channels, key_channels, value_channels, out_channels=None, num_heads=1,
                 att_norm_layer_type='IN'):
        super().__init__()
        if att_norm_layer_type != 'IN':
            raise NotImplementedError("liblavt_hip PWAM: att_norm_layer_type 'IN' only (reference default)")
        self.v_in_channels, self.l_in_channels = v_in_channels, l_in_channels
        self.key_channels, self.value_channels = key_channels, value_channels
        self.out_channels = out_channels or value_channels
        self.num_heads = num_heads
        if not (v_in_channels == key_channels == value_channels == self.out_channels):
            raise NotImplementedError("liblavt_hip PWAM: equal channel counts (as built by MMBasicLayer)")
        self.f_key = nn.Sequential(nn.Conv1d(l_in_channels, key_channels, 1))
        self.f_query = nn.Sequential(nn.Conv1d(v_in_channels, key_channels, 1), nn.InstanceNorm1d(key_channels))
        self.f_value = nn.Sequential(nn.Conv1d(l_in_channels, value_channels, 1))
        self.W = nn.Sequential(nn.Conv1d(value_channels, self.out_channels, 1), nn.InstanceNorm1d(self.out_channels))

    def rows(self, x2, B, T, lang, mul=None):
        """x2 [B*T, C] -> IN(W(attn)) (* mul) as [B*T, C]."""
        G = self.num_heads
        q = ops.instance_norm(ops.linear(x2, self.f_query[0].weight, self.f_query[0].bias), B, T)
        k, v, kv_sinks = lang.kv(self.f_key[0], self.f_value[0])
        o = ops.pwam_attention(q, k, v, lang.maskbias, B, T, lang.n_l, G, kv_sinks)
        return ops.instance_norm(ops.linear(o, self.W[0].weight, self.W[0].bias), B, T, mul=mul)

    def forward(self, x, l, l_mask):
        B, T, C = x.shape
        lang = _LangCtx.get(l, l_mask, x.dtype)
        return self.rows(x.reshape(B * T, C), B, T, lang).view(B, T, C)


class PWAM(nn.Module):
    """Pixel-word attention module.  Reference: lib/backbone.py:1238-1278."""

    def __init__(self, dim, v_in_channels, l_in_channels, key_channels, value_channels, num_heads=0, dropout=0.0,
                 attention=True, att_norm_layer_type='IN'):
        super().__init__()
        if not attention or dropout != 0.0:
            raise NotImplementedError("liblavt_hip PWAM: attention=True, dropout=0 (reference defaults)")
        self.vis_project = nn.Sequential(nn.Conv1d(dim, dim, 1), nn.GELU(), nn.Dropout(dropout))
        self.image_lang_att = SpatialImageLanguageAttention(v_in_channels, l_in_channels, key_channels, value_channels,
                                                            out_channels=value_channels, num_heads=max(num_heads, 1),
                                                            att_norm_layer_type=att_norm_layer_type)
        self.project_mm = nn.Sequential(nn.Conv1d(value_channels, value_channels, 1), nn.GELU(), nn.Dropout(dropout))

    def rows(self, x2, B, T, lang):
        vis = ops.linear(x2, self.vis_project[0].weight, self.vis_project[0].bias, act=ACT_GELU)
        mm = self.image_lang_att.rows(x2, B, T, lang, mul=vis)          # vis * IN(W(attn)) fused into the normalise pass
        return ops.linear(mm, self.project_mm[0].weight, self.project_mm[0].bias, act=ACT_GELU)

    def forward(self, x, l, l_mask):
        B, T, C = x.shape
        lang = _LangCtx.get(l, l_mask, x.dtype)
        return self.rows(x.reshape(B * T, C), B, T, lang).view(B, T, C)


class MMBasicLayer(nn.Module):
    """One stage: Swin blocks -> PWAM -> language gate -> PatchMerging.  Reference: lib/backbone.py:523-686."""

    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False, num_heads_fusion=1,
                 fusion_drop=0.0, args=None):
        super().__init__()
        self.window_size, self.shift_size, self.depth, self.dim = window_size, window_size // 2, depth, dim
        self.use_checkpoint = use_checkpoint
        self.version = getattr(args, "version", "default")
        self.fuse = getattr(args, "fuse", "default")
        self.hs = bool(getattr(args, "hs", False))
        self.lazy_pred = bool(getattr(args, "lazy_pred", False))
        for flag in ("bcam", "gacd", "efn"):
            if getattr(args, flag, False):
                raise NotImplementedError(f"--{flag} fusion is a comparison baseline outside the LAVT hot path")
        if self.fuse == "simple":
            raise NotImplementedError("--fuse simple (LangProject ablation) is outside the LAVT hot path")
        self.lg_act = getattr(args, "lg_act_layer", "tanh")
        if self.lg_act != "tanh":
            raise NotImplementedError("liblavt_hip language gate: tanh only (reference default)")
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, qk_scale,
                                 drop, attn_drop, drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path,
                                 norm_layer=norm_layer)
            for i in range(depth)])
        self.fusion = PWAM(dim, dim, 768, dim, dim, num_heads=num_heads_fusion, dropout=fusion_drop,
                           att_norm_layer_type=getattr(args, "att_norm_layer_type", "IN"))
        if self.version == "default":
            self.res_gate = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.ReLU(), nn.Linear(dim, dim, bias=False), nn.Tanh())
            nn.init.zeros_(self.res_gate[0].weight)
            nn.init.zeros_(self.res_gate[2].weight)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def never_used_parameters(self):
        """The last stage has no successor: its gated x is discarded and its gate never receives a gradient unless --hs returns the gated
        features (reference lib/backbone.py:669-686; the reference needs find_unused_parameters=True for it, train.py:592).  Read by
        lavt_hip.ddp.late_gradient_parameters."""
        if self.downsample is None and self.version == "default" and not self.hs:
            return list(self.res_gate.parameters())
        return []

    def forward(self, x, H, W, l, l_mask):
        B, L, C = x.shape
        for blk in self.blocks:
            blk.H, blk.W = H, W
            x = blk(x)
        x2 = xin = x.reshape(B * L, C)
        lang = _LangCtx.get(l, l_mask, x.dtype)
        with scope("pwam"):
            sila = self.fusion.image_lang_att
            if self.version == "default" and ops.pwam_fused_ok(x2, sila.num_heads):
                # PWAM + gate as one autograd node on the fused kernels (csrc/pwam.hip): 10 launches forward instead of ~15, ~17 backward instead of ~35
                k, v, kv_sinks = lang.kv(sila.f_key[0], sila.f_value[0])
                r, x2 = ops.pwam_gate(x2, k, v, lang.maskbias, kv_sinks, B, L, lang.n_l, self.fusion, self.res_gate)
            else:
                r = self.fusion.rows(x2, B, L, lang)
                if self.version == "default":
                    g = ops.linear(ops.linear(r, self.res_gate[0].weight, None, act=ACT_RELU), self.res_gate[2].weight, None)
                    x2 = ops.gate(x2, g, r)                               # x + tanh(g) * r
                elif self.version == "no_gate":
                    x2 = x2 + r
        feat = x2 if self.hs else (xin if self.lazy_pred else r)
        xg = x2.view(B, L, C)
        if self.downsample is not None:
            return feat.view(B, L, C), H, W, self.downsample(xg, H, W), (H + 1) // 2, (W + 1) // 2
        return feat.view(B, L, C), H, W, xg, H, W


class MultiModalSwinTransformer(nn.Module):
    """Reference: lib/backbone.py:334-520."""

    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1],
                 fusion_drop=0.0, args=None):
        super().__init__()
        if ape or drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("liblavt_hip backbone: ape=False, no dropout (LAVT factories never set them)")
        for flag in ("ytvos_2d_swin_3d_pwam", "ytvos_2d_swin_pwam"):
            if getattr(args, flag, False):
                raise NotImplementedError(f"--{flag} (2-D Swin on video ablation) is outside the LAVT hot path")
        self.pretrain_img_size, self.num_layers, self.embed_dim = pretrain_img_size, len(depths), embed_dim
        self.ape, self.patch_norm, self.out_indices, self.frozen_stages = ape, patch_norm, out_indices, frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(MMBasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint,
                num_heads_fusion=num_heads_fusion[i], fusion_drop=fusion_drop, args=args))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", norm_layer(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 2:
            for i in range(self.frozen_stages - 1):
                self.layers[i].eval()
                for p in self.layers[i].parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):
        """trunc_normal(0.02) on every nn.Linear (this also overwrites the zero-initialised gate, as in the
        reference: lib/backbone.py:472-476 vs :622-623), LayerNorm -> (1, 0); Conv1d keep PyTorch defaults."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if pretrained is not None and not isinstance(pretrained, str):
            raise TypeError('pretrained must be a str or None')
        self.apply(_init)
        if isinstance(pretrained, str):
            from lavt_hip.checkpoint import load_swin_checkpoint
            load_swin_checkpoint(self, pretrained)

    def _draw_drop_path(self, B, device):
        """All DropPath factors of one forward (two per block: attention branch, MLP branch) with 4 launches instead of 4 per draw."""
        dps = [blk.drop_path for layer in self.layers for blk in layer.blocks]
        live = [d for d in dps if self.training and d.drop_prob > 0.]
        if not live:
            return
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != device or keep.shape[0] != 2 * len(live):
            keep = torch.tensor([1.0 - d.drop_prob for d in live for _ in (0, 1)], dtype=torch.float32, device=device)[:, None]
            self._dp_keep = keep
        if os.environ.get("LAVT_DROPPATH_RNG", "device") == "torch":
            f = ops.droppath_factors(torch.rand(2 * len(live), B, device=device, dtype=torch.float32), keep)      # floor(keep + u) / keep, one launch
        else:
            f = ops.droppath_draw(keep, B)                                                                         # the same with the draw made in the kernel
        for i, d in enumerate(live):
            d._batched = [f[2 * i], f[2 * i + 1]]

    def forward(self, x, l, l_mask):
        dtype = compute_dtype()
        B = x.shape[0]
        self._draw_drop_path(B, x.device)
        lang = _LangCtx.get(l, l_mask, dtype)
        lang.set_plan([(layer.fusion.image_lang_att.f_key[0], layer.fusion.image_lang_att.f_value[0]) for layer in self.layers])
        t, Wh, Ww = self.patch_embed.tokens(x, dtype)
        t = t.view(B, Wh * Ww, self.embed_dim)
        outs = []
        for i, layer in enumerate(self.layers):
            f, H, W, t, Wh, Ww = layer(t, Wh, Ww, l, l_mask)
            if i in self.out_indices:
                nl = getattr(self, f"norm{i}")
                C = self.num_features[i]
                fn = ops.layer_norm(f.reshape(B * H * W, C), nl.weight, nl.bias, nl.eps)
                outs.append(fn.view(B, H, W, C).permute(0, 3, 1, 2))          # NCHW-shaped view of NHWC memory
        lang.set_plan(None)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self
