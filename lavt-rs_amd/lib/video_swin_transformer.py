"""MI355X-native multi-modal Video-Swin backbone of LAVT (video path).

Drop-in for the reference's lib/video_swin_transformer.py on the hot path: same class names, constructor arguments,
sub-module names (=> identical state-dict keys) and `MultiModalSwinTransformer3D.forward(x, l, l_mask)` contract
((B, 3, T, H, W) clip -> four (B*T, C_i, H_i, W_i) maps, reference :854-881); every computation is a liblavt_hip
kernel launched through lavt_hip.ops.

Data layout: tokens stay [B*D*H*W, C] (NDHWC) through the whole backbone.  The 3-D pad -> roll -> window_partition ...
window_reverse -> roll -> crop sequence (:230-262) is a row map on the qkv / proj GEMMs; the dense 0/-100 mask of
compute_mask (:315-328) is an int8 region table; Conv3d of SepTPWAM is an implicit GEMM over NDHWC rows with a
kd x kh x kw tap gather.  Windows of up to 160 tokens run in the fused attention kernels, larger ones (8x7x7 = 392,
8x12x12 = 1152) through the composed GEMM -> softmax -> GEMM path.

In scope: the default per-stage PWAM and the README training recipe
`--sep_t_pwam --conv3d_kernel_size_t 3-3-3 --conv3d_kernel_size_s 1-1-1 --w_t3x3_s1x1 --mm_t3x3_s1x1` (SepTPWAM).
The other fusion ablations (TSPWAM, TPWAM, TPWAMComp, SepTPWAMInner, SeqTPWAM, SepSeqTPWAM*, LangProject) raise
NotImplementedError.
"""
import os

import torch
import torch.nn as nn

from lavt_hip import ops, rowmaps
from lavt_hip._capi import ACT_GELU, ACT_NONE, ACT_RELU
from lavt_hip.runtime import compute_dtype

from .backbone import PWAM, DropPath, Mlp, _LangCtx

sr_ratio = [1, 1, 1, 1]     # module-level name the reference reads while building stages (its own file never defines it)


def get_window_size(x_size, window_size, shift_size=None):
    """Reference :70-83."""
    return rowmaps.clip_window(tuple(x_size), tuple(window_size), None if shift_size is None else tuple(shift_size))


def _parse3(s):
    return tuple(int(a) for a in s.split('-'))


class WindowAttention3D(nn.Module):
    """Reference :86-168.  Parameter container + forward on already partitioned windows."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        if dim // num_heads != 32 or qk_scale is not None or attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("liblavt_hip WindowAttention3D: head_dim 32, default scale, no dropout")
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        d, h, w = (a.reshape(-1) for a in torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij"))
        idx = ((d[:, None] - d[None] + wd - 1) * (2 * wh - 1) + (h[:, None] - h[None] + wh - 1)) * (2 * ww - 1) + (w[:, None] - w[None] + ww - 1)
        self.register_buffer("relative_position_index", idx)                    # state-dict compatibility only
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, mask=None):
        """x (num_windows*B, N, C); mask: None or the int8 region table of rowmaps.region_ids3d.  With N smaller than the
        full window the bias is the top-left N x N block of the full index matrix, as in the reference (:150)."""
        if mask is not None and mask.dtype != torch.int8:
            raise TypeError("pass the int8 region table (rowmaps.region_ids3d), not a dense float mask")
        B_, N, C = x.shape
        qkv = ops.linear(x.reshape(B_ * N, C), self.qkv.weight, self.qkv.bias)
        a = ops.window_attention(qkv, self.relative_position_bias_table, mask, self.window_size, self.num_heads, N=N)
        return ops.linear(a, self.proj.weight, self.proj.bias).view(B_, N, C)


class SwinTransformerBlock3D(nn.Module):
    """Reference :171-273.  x: (B, D, H, W, C)."""

    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_checkpoint=False):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, tuple(window_size), tuple(shift_size)
        self.mlp_ratio, self.use_checkpoint = mlp_ratio, use_checkpoint
        assert all(0 <= s < w for s, w in zip(self.shift_size, self.window_size)), "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention3D(dim, self.window_size, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, mask_matrix=None):
        """mask_matrix is accepted for signature compatibility and ignored: the region table is derived from the shapes."""
        B, D, H, W, C = x.shape
        dev = x.device
        win, shift = get_window_size((D, H, W), self.window_size, self.shift_size)
        L = D * H * W
        x2 = x.reshape(B * L, C)
        wmap = rowmaps.window_map3d(B, D, H, W, win, shift, dev)
        region = rowmaps.region_ids3d(D, H, W, win, shift, dev) if any(s > 0 for s in shift) else None
        M = wmap.numel()
        a = self.attn
        xn, x2 = ops.layer_norm_res(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps)     # x2: alias for the residual branch
        qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias, in_map=wmap, rows=M)
        o = ops.window_attention(qkv, a.relative_position_bias_table, region, self.window_size, self.num_heads, N=win[0] * win[1] * win[2])
        f1 = self.drop_path.factors(B, dev)
        dpv = 1.0 / (1.0 - self.drop_path.drop_prob) if self.drop_path.drop_prob < 1.0 else 0.0
        x2 = ops.linear(o, a.proj.weight, a.proj.bias, residual=x2, out_map=wmap, rows=M, out_rows=B * L,
                        row_scale=f1, row_scale_div=M // B, row_scale_value=dpv)
        f2 = self.drop_path.factors(B, dev)
        h, x2 = ops.layer_norm_res(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x2 = self.mlp(h, residual=x2, row_scale=f2, row_scale_div=L, row_scale_value=dpv)
        return x2.view(B, D, H, W, C)


class PatchMerging(nn.Module):
    """Reference :276-311: spatial 2x2 merge per frame, x (B, D, H, W, C) -> (B, D, ceil(H/2), ceil(W/2), 2C)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        B, D, H, W, C = x.shape
        g = rowmaps.merge_map(B * D, H, W, x.device)               # frames are independent images for the merge
        z = ops.layer_norm(x.reshape(B * D * H * W, C), self.norm.weight, self.norm.bias, self.norm.eps, gather=g)
        y = ops.linear(z, self.reduction.weight, None)
        return y.view(B, D, (H + 1) // 2, (W + 1) // 2, 2 * C)


class PatchEmbed3D(nn.Module):
    """Reference :594-634 with the LAVT patch size (1, 4, 4): a per-frame 4x4/4 convolution."""

    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        if tuple(patch_size) != (1, 4, 4) or in_chans != 3:
            raise NotImplementedError("liblavt_hip PatchEmbed3D: (1, 4, 4) patches of 3-channel clips (what lavt_video builds)")
        self.patch_size, self.in_chans, self.embed_dim = tuple(patch_size), in_chans, embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def tokens(self, x, dtype):
        """x (B, 3, T, H, W) -> rows [B*T*H4*W4, C0], (T, H4, W4)"""
        B, _, T, H, W = x.shape
        frames = x.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)              # a view again when x came from LAVTVideo's permute
        t = ops.patch_embed(frames, self.proj.weight, self.proj.bias, dtype)
        if self.norm is not None:
            t = ops.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps)
        return t, T, (H + 3) // 4, (W + 3) // 4

    def forward(self, x):
        t, T, Wh, Ww = self.tokens(x, compute_dtype())
        return t.view(x.shape[0], T, Wh, Ww, self.embed_dim).permute(0, 4, 1, 2, 3)


def _3d_kernel_size_to_padding_size(ks):
    return None if ks is None else tuple(k // 2 for k in ks)


class SepTPWAM(nn.Module):
    """Separated temporal / spatial pixel-word attention (reference :1300-1584), README recipe only:
    t and s branches for the visual projection, the query, W and project_mm, each pair summed."""

    def __init__(self, dim, v_in_channels, l_in_channels, key_channels, value_channels, num_heads=0, dropout=0.0,
                 conv3d_kernel_size_t=(3, 1, 1), conv3d_kernel_size_s=(1, 1, 1), w_3x3=False, mm_3x3=False, w_3=False, mm_3=False,
                 sum_3_kernel_size=None, cat_reduce_kernel_size=None, w_t3x3_s1x1=None, mm_t3x3_s1x1=None, args=None):
        super().__init__()
        gates = [getattr(args, n, False) for n in ("s_tanh_plus_1_gate_1_q", "s_tanh_plus_1_gate_1_v", "t_tanh_plus_1_gate_1_q", "t_tanh_plus_1_gate_1_v")]
        if (w_3x3 or mm_3x3 or w_3 or mm_3 or sum_3_kernel_size or cat_reduce_kernel_size or any(gates) or dropout != 0.0
                or not (w_t3x3_s1x1 and mm_t3x3_s1x1)):
            raise NotImplementedError("liblavt_hip SepTPWAM: the README recipe (--w_t3x3_s1x1 --mm_t3x3_s1x1, no extra gates / fuse convs / dropout)")
        if not (dim == v_in_channels == key_channels == value_channels):
            raise NotImplementedError("liblavt_hip SepTPWAM: equal channel counts (as built by MMBasicLayer)")
        kt, ks = tuple(conv3d_kernel_size_t), tuple(conv3d_kernel_size_s)
        if any(k not in (1, 3) for k in kt + ks):
            raise NotImplementedError("liblavt_hip SepTPWAM: kernel sizes 1 or 3 per axis")
        self.num_heads = max(num_heads, 1)
        self.w_t3x3_s1x1, self.mm_t3x3_s1x1 = w_t3x3_s1x1, mm_t3x3_s1x1
        pt, ps = _3d_kernel_size_to_padding_size(kt), _3d_kernel_size_to_padding_size(ks)

        def conv(k, p):
            return nn.Conv3d(dim, dim, kernel_size=k, stride=1, padding=p)
        self.temporal_vis_project = nn.Sequential(conv(kt, pt), nn.GELU(), nn.Dropout(dropout))
        self.spatial_vis_project = nn.Sequential(conv(ks, ps), nn.GELU(), nn.Dropout(dropout))
        self.f_query_t = nn.Sequential(conv(kt, pt), nn.InstanceNorm3d(dim))
        self.f_query_s = nn.Sequential(conv(ks, ps), nn.InstanceNorm3d(dim))
        self.f_key = nn.Sequential(nn.Conv1d(l_in_channels, key_channels, 1))
        self.f_value = nn.Sequential(nn.Conv1d(l_in_channels, value_channels, 1))
        self.W_t = nn.Sequential(conv(kt, pt), nn.InstanceNorm3d(dim))                          # reference :1435-1442: t kernel / 1x1x1
        self.W_s = nn.Sequential(conv((1, 1, 1), (0, 0, 0)), nn.InstanceNorm3d(dim))
        self.project_mm_t = nn.Sequential(conv(kt, pt), nn.GELU(), nn.Dropout(dropout))
        self.project_mm_s = nn.Sequential(conv((1, 1, 1), (0, 0, 0)), nn.GELU(), nn.Dropout(dropout))

    def rows(self, x2, B, D, H, W, lang):
        """x2 [B*D*H*W, C] -> [B*D*H*W, C]"""
        T = D * H * W

        def c3(x, seq, act=ACT_NONE, residual=None):
            m = seq[0]
            if tuple(m.kernel_size) == (1, 1, 1):
                return ops.linear(x, m.weight, m.bias, act=act, residual=residual)
            y = ops.conv3d(x, m.weight, m.bias, B, D, H, W, act=act)
            return y if residual is None else y + residual
        vis = c3(x2, self.spatial_vis_project, ACT_GELU, residual=c3(x2, self.temporal_vis_project, ACT_GELU))
        q = ops.instance_norm(c3(x2, self.f_query_t), B, T) + ops.instance_norm(c3(x2, self.f_query_s), B, T)
        k, v, kv_sinks = lang.kv(self.f_key[0], self.f_value[0])
        o = ops.pwam_attention(q, k, v, lang.maskbias, B, T, lang.n_l, self.num_heads, kv_sinks)
        mm = ops.instance_norm(c3(o, self.W_t), B, T, mul=vis) + ops.instance_norm(c3(o, self.W_s), B, T, mul=vis)      # vis * (W_t + W_s)
        return c3(mm, self.project_mm_s, ACT_GELU, residual=c3(mm, self.project_mm_t, ACT_GELU))

    def forward(self, x, l, l_mask):
        """x (B, D, H, W, C) -> (B, D*H*W, C)"""
        B, D, H, W, C = x.shape
        lang = _LangCtx.get(l, l_mask, x.dtype)
        return self.rows(x.reshape(B * D * H * W, C), B, D, H, W, lang).view(B, D * H * W, C)


_UNSUPPORTED_FUSIONS = ("ts_pwam", "t_pwam", "t_pwam_comp", "seq_t_pwam", "sep_t_pwam_inner", "sep_seq_t_pwam", "sep_seq_t_pwam_inner")


class MMBasicLayer(nn.Module):
    """One video stage: Video-Swin blocks -> PWAM | SepTPWAM -> language gate -> per-frame PatchMerging (reference :331-591).
    forward(x (B, C, D, H, W)-shaped) -> (stage feature, next-stage input), both (B, C', D, H', W')-shaped views of NDHWC memory."""

    def __init__(self, dim, depth, num_heads, window_size=(1, 7, 7), mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False, num_heads_fusion=1, fusion_drop=0.0,
                 sr_ratio=1, args=None):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in self.window_size)
        self.depth, self.dim, self.use_checkpoint = depth, dim, use_checkpoint
        self.version = getattr(args, "version", "default")
        self.fuse = getattr(args, "fuse", "default")
        self.hs = bool(getattr(args, "hs", False))
        self.lazy_pred = bool(getattr(args, "lazy_pred", False))
        self.is_last_layer = num_heads in (24, 32)
        self.sep_t_pwam = bool(getattr(args, "sep_t_pwam", False))
        for flag in _UNSUPPORTED_FUSIONS:
            if getattr(args, flag, False):
                raise NotImplementedError(f"--{flag} fusion ablation is outside the LAVT video hot path (PWAM and --sep_t_pwam are built)")
        if self.fuse == "simple":
            raise NotImplementedError("--fuse simple (LangProject ablation) is outside the LAVT hot path")
        if sr_ratio != 1:
            raise NotImplementedError("liblavt_hip video stage: sr_ratio 1")
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, self.window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio, qkv_bias,
                                   qk_scale, drop, attn_drop, drop_path[i] if isinstance(drop_path, (list, tuple)) else drop_path,
                                   norm_layer=norm_layer, use_checkpoint=use_checkpoint)
            for i in range(depth)])
        if self.sep_t_pwam:
            sum3 = getattr(args, "sept_sum_3_kernel_size", None)
            cat3 = getattr(args, "sept_cat_reduce_kernel_size", None)
            self.fusion = SepTPWAM(dim, dim, 768, dim, dim, num_heads=num_heads_fusion, dropout=fusion_drop,
                                   conv3d_kernel_size_t=_parse3(args.conv3d_kernel_size_t), conv3d_kernel_size_s=_parse3(args.conv3d_kernel_size_s),
                                   w_3x3=getattr(args, "w_3x3", False), mm_3x3=getattr(args, "mm_3x3", False), w_3=getattr(args, "w_3", False),
                                   mm_3=getattr(args, "mm_3", False), sum_3_kernel_size=_parse3(sum3) if sum3 else None,
                                   cat_reduce_kernel_size=_parse3(cat3) if cat3 else None,
                                   w_t3x3_s1x1=getattr(args, "w_t3x3_s1x1", False), mm_t3x3_s1x1=getattr(args, "mm_t3x3_s1x1", False), args=args)
        else:
            self.fusion = PWAM(dim, dim, 768, dim, dim, num_heads=num_heads_fusion, dropout=fusion_drop)
        if self.version == "default" and not (self.is_last_layer and self.use_checkpoint):
            self.res_gate = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.ReLU(), nn.Linear(dim, dim, bias=False), nn.Tanh())
            nn.init.zeros_(self.res_gate[0].weight)
            nn.init.zeros_(self.res_gate[2].weight)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def never_used_parameters(self):
        """the last stage's gate (when it exists) feeds only the discarded gated x: no gradient, see lib/backbone.py"""
        if self.downsample is None and hasattr(self, "res_gate") and not self.hs:
            return list(self.res_gate.parameters())
        return []

    def rows(self, x, l, l_mask):
        """x (B, D, H, W, C) NDHWC -> (feature (B, D, H, W, C), next (B, D, H', W', C'))"""
        B, D, H, W, C = x.shape
        for blk in self.blocks:
            x = blk(x)
        L = D * H * W
        x2 = x.reshape(B * L, C)
        lang = _LangCtx.get(l, l_mask, x.dtype)
        gated = self.version == "default" and (not self.use_checkpoint or not self.is_last_layer)
        fused = gated and not self.sep_t_pwam and ops.pwam_fused_ok(x2, self.fusion.image_lang_att.num_heads)
        if fused:                                                     # PWAM + gate as one autograd node (csrc/pwam.hip)
            sila = self.fusion.image_lang_att
            k, v, kv_sinks = lang.kv(sila.f_key[0], sila.f_value[0])
            r, xg = ops.pwam_gate(x2, k, v, lang.maskbias, kv_sinks, B, L, lang.n_l, self.fusion, self.res_gate)
        else:
            r = self.fusion.rows(x2, B, D, H, W, lang) if self.sep_t_pwam else self.fusion.rows(x2, B, L, lang)
            xg = x2
        if not fused and gated:
            g = ops.linear(ops.linear(r, self.res_gate[0].weight, None, act=ACT_RELU), self.res_gate[2].weight, None)
            xg = ops.gate(x2, g, r)                                   # x + tanh(g) * r
        elif not fused and self.version == "no_gate":
            xg = x2 + r
        feat = xg if self.hs else (x2 if self.lazy_pred else r)
        xg = xg.view(B, D, H, W, C)
        nxt = self.downsample(xg) if self.downsample is not None else xg
        return feat.view(B, D, H, W, C), nxt

    def forward(self, x, l, l_mask):
        f, nxt = self.rows(x.permute(0, 2, 3, 4, 1), l, l_mask)
        return f.permute(0, 4, 1, 2, 3), nxt.permute(0, 4, 1, 2, 3)


class MultiModalSwinTransformer3D(nn.Module):
    """Reference :637-886."""

    def __init__(self, pretrained=None, pretrained2d=False, patch_size=(4, 4, 4), in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=(2, 7, 7), mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, patch_norm=False, out_indices=(0, 1, 2, 3),
                 frozen_stages=-1, use_checkpoint=False, num_heads_fusion=[1, 1, 1, 1], fusion_drop=0.0, args=None):
        super().__init__()
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("liblavt_hip video backbone: no dropout (lavt_video never sets it)")
        self.pretrained, self.pretrained2d = pretrained, pretrained2d
        self.num_layers, self.embed_dim, self.patch_norm = len(depths), embed_dim, patch_norm
        self.out_indices, self.frozen_stages = out_indices, frozen_stages
        self.window_size, self.patch_size = tuple(window_size), tuple(patch_size)
        self.patch_embed = PatchEmbed3D(patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(MMBasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=self.window_size, mlp_ratio=mlp_ratio,
                qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint,
                num_heads_fusion=num_heads_fusion[i], fusion_drop=fusion_drop, sr_ratio=sr_ratio[i], args=args))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", norm_layer(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1:
            for i in range(self.frozen_stages):
                self.layers[i].eval()
                for p in self.layers[i].parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):
        """trunc_normal(0.02) on nn.Linear (this also overwrites the zero-initialised gates, as in the reference), LayerNorm (1, 0)."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if pretrained:
            self.pretrained = pretrained
        if self.pretrained is not None and not isinstance(self.pretrained, str):
            raise TypeError('pretrained must be a str or None')
        self.apply(_init)
        if isinstance(self.pretrained, str):
            from lavt_hip.checkpoint import load_video_swin_checkpoint
            load_video_swin_checkpoint(self, self.pretrained, inflate_2d=self.pretrained2d)

    def _draw_drop_path(self, B, device):
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
        """x (B, 3, T, H, W) -> tuple of (B*T, C_i, H_i, W_i)-shaped maps (NHWC memory)"""
        dtype = compute_dtype()
        B = x.shape[0]
        self._draw_drop_path(B, x.device)
        lang = _LangCtx.get(l, l_mask, dtype)
        fusions = [layer.fusion if layer.sep_t_pwam else layer.fusion.image_lang_att for layer in self.layers]
        lang.set_plan([(f.f_key[0], f.f_value[0]) for f in fusions])
        t, T, Wh, Ww = self.patch_embed.tokens(x, dtype)
        t = t.view(B, T, Wh, Ww, self.embed_dim)
        outs = []
        for i, layer in enumerate(self.layers):
            f, t = layer.rows(t, l, l_mask)
            if i in self.out_indices:
                nl = getattr(self, f"norm{i}")
                _, D, H, W, C = f.shape
                fn = ops.layer_norm(f.reshape(B * D * H * W, C), nl.weight, nl.bias, nl.eps)
                outs.append(fn.view(B * D, H, W, C).permute(0, 3, 1, 2))
        lang.set_plan(None)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self
