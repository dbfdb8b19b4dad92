"""Top-level LAVT modules with the reference's forward signatures (lib/_utils.py:10-108).

    LAVT(backbone, classifier).forward(x, l_feats, l_mask)        -> (B, 2, H, W) fp32 logits
    LAVTOne(backbone, classifier, args).forward(x, text, l_mask)  (BERT inside)

The backbone / decoder arithmetic is liblavt_hip; the final align_corners bilinear upsample writes
the NCHW fp32 logits directly.  BERT (the absent ./bert package of the reference) is not part of the
hot path: LAVTOne uses `bert.modeling_bert.BertModel` when importable, else transformers' BertModel.
"""
import os

import torch
from torch import nn

from lavt_hip import ops
from lavt_hip.runtime import compute_dtype
from .mask_predictor import nchw_rows


def _upsample_logits(y, size):
    """y: (B, 2, h, w)-shaped decoder output -> (B, 2, H, W) fp32, bilinear align_corners=True (lib/_utils.py:21)."""
    B, _, h, w = y.shape
    return ops.logits_upsample(nchw_rows(y, y.dtype), B, h, w, int(size[0]), int(size[1]))


def fused_loss(y, target, weight=(0.9, 1.1)):
    """The caller's `criterion(model(...), target)` (train.py:221-224; losses.py:7-11) on the LOW-resolution decoder output y
    ((B, 2, h, w)-shaped): bilinear upsample to target.shape[-2:] + weighted cross-entropy + I/U counts in one kernel pair.
    -> (loss, stats[loss, sum of weights, I, U])"""
    B, _, h, w = y.shape
    return ops.upsample_cross_entropy(nchw_rows(y, y.dtype), target, B, h, w, int(target.shape[-2]), int(target.shape[-1]), weight)


def fused_dice_loss(y, target):
    """`MultiClassDiceLoss()(model(...), target)` (train.py:703-704; losses.py:38-77) on the LOW-resolution decoder output y: bilinear upsample
    + softmax + per-sample Dice sums in one kernel pair.  -> (loss, stats)"""
    B, _, h, w = y.shape
    return ops.upsample_dice_loss(nchw_rows(y, y.dtype), target, B, h, w, int(target.shape[-2]), int(target.shape[-1]))


class _LAVTSimpleDecode(nn.Module):
    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier

    def forward_lowres(self, x, l_feats, l_mask):
        """decoder output before the final upsample, (B, 2, H/4, W/4)-shaped: feed it to `fused_loss`"""
        x_c1, x_c2, x_c3, x_c4 = self.backbone(x, l_feats, l_mask)
        return self.classifier(x_c4, x_c3, x_c2, x_c1)

    def forward(self, x, l_feats, l_mask):
        return _upsample_logits(self.forward_lowres(x, l_feats, l_mask), x.shape[-2:])


class LAVT(_LAVTSimpleDecode):
    pass


def _build_text_encoder(args):
    """`BertModel.from_pretrained(args.ck_bert)` of the reference (lib/_utils.py:38-40), on the liblavt_hip encoder (bert/modeling_bert.py).
    `args.ck_bert` must be a directory with config.json + weights: there is no hub download here, and a missing checkpoint raises instead of
    silently training / evaluating with an untrained text encoder.  Random bert-base-uncased geometry is available only on request:
    `args.bert_random_init = True` (benchmarks and tests on synthetic data) or LAVT_BERT_RANDOM_INIT=1."""
    from bert.modeling_bert import BertConfig, BertModel
    ck = getattr(args, "ck_bert", "bert-base-uncased")
    if os.path.isdir(str(ck)):
        enc = BertModel.from_pretrained(ck)
    elif getattr(args, "bert_random_init", False) or os.environ.get("LAVT_BERT_RANDOM_INIT", "0") == "1":
        import warnings
        warnings.warn(f"text encoder: ck_bert={ck!r} is not a checkpoint directory; building a RANDOMLY INITIALISED bert-base-uncased on request")
        enc = BertModel(BertConfig())
    else:
        raise FileNotFoundError(f"text encoder: ck_bert={ck!r} is not a directory with config.json + pytorch_model.bin / model.safetensors "
                                "(no hub download here); pass args.bert_random_init=True for a randomly initialised encoder")
    enc.pooler = None
    return enc


class _LAVTOneSimpleDecode(nn.Module):
    def __init__(self, backbone, classifier, args):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.text_encoder = _build_text_encoder(args)
        self.lazy_pred = bool(getattr(args, "lazy_pred", False))

    def forward_lowres(self, x, text, l_mask):
        """token ids (B, N_l) + attention mask (B, N_l) -> decoder logits at 1/4 resolution (what the fused upsample + CE kernel consumes)"""
        l_feats = self.text_encoder(text, attention_mask=l_mask)[0].permute(0, 2, 1)      # (B, 768, N_l)
        l_mask = l_mask.unsqueeze(dim=-1)
        features = self.backbone(x, l_feats, l_mask)
        if self.lazy_pred:
            x_c1, (x_c2, x_c3, x_c4) = None, features
        else:
            x_c1, x_c2, x_c3, x_c4 = features
        return self.classifier(x_c4, x_c3, x_c2, x_c1)

    def forward(self, x, text, l_mask):
        return _upsample_logits(self.forward_lowres(x, text, l_mask), x.shape[-2:])


class LAVTOne(_LAVTOneSimpleDecode):
    pass


class _LAVTVideoSimpleDecode(nn.Module):
    """Reference lib/_utils.py:76-108: clip (B, T, 3, H, W) + token ids (B, N_l) + attention mask (B, N_l) -> (B*T, 2, H, W) logits.
    BERT runs inside (it is outside the hot path, as for LAVTOne); `forward_backbone` is the hot path proper on language features."""

    def __init__(self, backbone, classifier, args):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.text_encoder = _build_text_encoder(args)
        self.lazy_pred = bool(getattr(args, "lazy_pred", False))
        self.seg_last = bool(getattr(args, "seg_last", False))

    def forward_backbone(self, x, l_feats, l_mask):
        """x (B, T, 3, H, W); l_feats (B, 768, N_l); l_mask (B, N_l, 1)"""
        input_shape = x.shape[-2:]
        features = self.backbone(x.permute(0, 2, 1, 3, 4), l_feats, l_mask)          # (B, 3, T, H, W) view; the patch embed reads frames
        if self.lazy_pred:
            x_c1, (x_c2, x_c3, x_c4) = None, features
        else:
            x_c1, x_c2, x_c3, x_c4 = features
        y = self.classifier(x_c4, x_c3, x_c2, x_c1)
        if self.seg_last:
            return y
        return _upsample_logits(y, input_shape)

    def forward(self, x, text, l_mask):
        l_feats = self.text_encoder(text, attention_mask=l_mask)[0].permute(0, 2, 1)      # (B, 768, N_l)
        return self.forward_backbone(x, l_feats, l_mask.unsqueeze(dim=-1))

    def forward_feats(self, x, text, l_mask):
        """Reference lib/_utils.py:110-131: -> (logits (B*T, 2, H, W) fp32, [x_c4, level-4, level-3, level-2 decoder features])."""
        input_shape = x.shape[-2:]
        l_feats = self.text_encoder(text, attention_mask=l_mask)[0].permute(0, 2, 1)
        features = self.backbone(x.permute(0, 2, 1, 3, 4), l_feats, l_mask.unsqueeze(dim=-1))
        if self.lazy_pred:
            x_c1, (x_c2, x_c3, x_c4) = None, features
        else:
            x_c1, x_c2, x_c3, x_c4 = features
        y, feats = self.classifier.forward_feats(x_c4, x_c3, x_c2, x_c1)
        return _upsample_logits(y, input_shape), feats

    def load_from_pretrained2d_lavt_weights(self, pretrained):
        """Reference lib/_utils.py:133-182 (called by train.py:575-576): released 2-D LAVT weights -> this video model (patch-embed
        `unsqueeze(2)`, bias tables bicubic-resized then repeated 2*Wd-1 times), non-strict load."""
        from lavt_hip.checkpoint import load_lavt2d_into_video
        missing, unexpected = load_lavt2d_into_video(self, pretrained, drop_fusion=False)
        print(f"=> loaded successfully '{pretrained}' (missing {len(missing)}, unexpected {len(unexpected)})")
        ops.weights.invalidate()
        return missing, unexpected

    def load_from_pretrained2d_lavt_weights_into_a_3d_model(self, pretrained):
        """Reference lib/_utils.py:184-238 (train.py:577-578): as above, but the '.fusion' tensors (2-D PWAM) are dropped."""
        from lavt_hip.checkpoint import load_lavt2d_into_video
        missing, unexpected = load_lavt2d_into_video(self, pretrained, drop_fusion=True)
        print(f"=> loaded successfully '{pretrained}' (missing {len(missing)}, unexpected {len(unexpected)})")
        ops.weights.invalidate()
        return missing, unexpected


class LAVTVideo(_LAVTVideoSimpleDecode):
    pass
