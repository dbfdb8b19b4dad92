"""MI355X-native top-down mask decoder.  Drop-in for lib/mask_predictor.py:SimpleDecoding of the
reference (same constructor, sub-module names / state-dict keys, forward(x_c4, x_c3, x_c2, x_c1)).

nn.Conv2d / nn.BatchNorm2d objects only hold parameters and running statistics; the arithmetic is
the implicit-GEMM 3x3 convolution of liblavt_hip (NHWC, concat of [top-down, skip] read through
two pointers -- torch.cat never materialises), fused BatchNorm(+ReLU) passes and a bilinear kernel.
`torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` is honoured: a converted bn module makes the
statistics all-reduce over its process group (train.py:589 of the reference).
Out of scope: `interpolate_before_seg` / `seg_last` branches, LTSDecoding, ASPP.
"""
import torch
from torch import nn

from lavt_hip import ops
from lavt_hip.runtime import compute_dtype


def nchw_rows(x, dtype):
    """(B,C,H,W)-shaped tensor -> NHWC rows [B*H*W, C] in `dtype`; zero-copy when x is a channels-last view."""
    B, C, H, W = x.shape
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous() and x.dtype == dtype:
        return xp.reshape(B * H * W, C)
    return ops.transpose_last2(x.reshape(B, C, H * W), dtype).view(B * H * W, C)


class SimpleDecoding(nn.Module):
    def __init__(self, c4_dims, args, factor=2):
        super().__init__()
        self.lazy_pred = bool(getattr(args, "lazy_pred", False))
        if getattr(args, "interpolate_before_seg", False) or getattr(args, "seg_last", False):
            raise NotImplementedError("interpolate_before_seg / seg_last decoder branches are outside the LAVT hot path")
        hidden = c4_dims // factor
        c3, c2, c1 = c4_dims // factor, c4_dims // factor ** 2, c4_dims // factor ** 3
        self.hidden_size = hidden

        def block(tag, cin):
            setattr(self, f"conv1_{tag}", nn.Conv2d(cin, hidden, 3, padding=1, bias=False))
            setattr(self, f"bn1_{tag}", nn.BatchNorm2d(hidden))
            setattr(self, f"relu1_{tag}", nn.ReLU())
            setattr(self, f"conv2_{tag}", nn.Conv2d(hidden, hidden, 3, padding=1, bias=False))
            setattr(self, f"bn2_{tag}", nn.BatchNorm2d(hidden))
            setattr(self, f"relu2_{tag}", nn.ReLU())
        block(4, c4_dims + c3)
        block(3, hidden + c2)
        if not self.lazy_pred:
            block(2, hidden + c1)
        self.conv1_1 = nn.Conv2d(hidden, 2, 1)

    def _level(self, tag, x, xhw, skip, B, dtype):
        """x: rows at resolution xhw; skip: NCHW-shaped feature.  Returns rows at the skip's resolution."""
        H, W = skip.shape[-2:]
        w1, w2 = getattr(self, f"conv1_{tag}").weight, getattr(self, f"conv2_{tag}").weight
        # fp8 (configs[4]): the producers of the convolutions' inputs write the e4m3 twins / record the gradient |max| themselves (None = that convolution is bf16)
        M, C1, hid = B * H * W, x.shape[1], w2.shape[0]
        s1, s2 = ops.fp8_act_site(w1, M, C1, skip.shape[1]), ops.fp8_act_site(w2, M, hid)
        d1, d2 = ops.fp8_dy_site(w1, M, C1), ops.fp8_dy_site(w2, M, hid)
        if xhw[0] < H or xhw[1] < W:
            x = ops.bilinear(x, B, xhw[0], xhw[1], H, W, fp8_site=s1)
        elif xhw != (H, W):
            raise ValueError("decoder: top-down map larger than the skip feature")
        x = ops.conv3x3(x, nchw_rows(skip, dtype), w1, B, H, W)
        x = ops.batch_norm_relu(x, getattr(self, f"bn1_{tag}"), track=False, fp8_site=s2, fp8_dy_site=d1)          # (num_batches_tracked: one launch for all layers, _run_scoped)
        x = ops.conv3x3(x, None, w2, B, H, W)
        x = ops.batch_norm_relu(x, getattr(self, f"bn2_{tag}"), track=False, fp8_dy_site=d2)
        return x, (H, W)

    def _run(self, x_c4, x_c3, x_c2, x_c1):
        from lavt_hip._capi import scope
        with scope("decoder"):
            return self._run_scoped(x_c4, x_c3, x_c2, x_c1)

    def _run_scoped(self, x_c4, x_c3, x_c2, x_c1):
        dtype = compute_dtype()
        B = x_c4.shape[0]
        feats = []
        x, hw = nchw_rows(x_c4, dtype), tuple(x_c4.shape[-2:])
        x, hw = self._level(4, x, hw, x_c3, B, dtype)
        feats.append((x, hw))
        x, hw = self._level(3, x, hw, x_c2, B, dtype)
        feats.append((x, hw))
        if not self.lazy_pred:
            x, hw = self._level(2, x, hw, x_c1, B, dtype)
            feats.append((x, hw))
        ops.bn_count_batches([getattr(self, f"bn{i}_{tag}") for tag in ((4, 3) if self.lazy_pred else (4, 3, 2)) for i in (1, 2)])
        y = ops.cls_head(x, self.conv1_1.weight, self.conv1_1.bias)
        as_nchw = lambda r, s: r.view(B, s[0], s[1], r.shape[1]).permute(0, 3, 1, 2)     # noqa: E731
        return as_nchw(y, hw), [as_nchw(f, s) for f, s in feats]

    def forward(self, x_c4, x_c3, x_c2, x_c1):
        return self._run(x_c4, x_c3, x_c2, x_c1)[0]

    def forward_feats(self, x_c4, x_c3, x_c2, x_c1):
        y, feats = self._run(x_c4, x_c3, x_c2, x_c1)
        return y, [x_c4] + feats
