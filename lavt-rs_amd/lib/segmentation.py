"""Model factories with the reference's interface: `segmentation.__dict__[args.model](pretrained, args)`
(train.py:572, test.py:273 of the reference; factories lib/segmentation.py:14-148).

Differences from the shipped reference, on purpose: `lavt` forwards `args` to the backbone (the
reference forgets to and cannot construct).  `args` may omit any attribute; defaults
are the reference's argparse defaults.
"""
from .backbone import MultiModalSwinTransformer
from .mask_predictor import SimpleDecoding
from ._utils import LAVT, LAVTOne, LAVTVideo
from .video_swin_transformer import MultiModalSwinTransformer3D

__all__ = ['lavt', 'lavt_one', 'lavt_video']

SWIN_VARIANTS = {          # lib/segmentation.py:16-31
    'tiny': (96, [2, 2, 6, 2], [3, 6, 12, 24]),
    'small': (96, [2, 2, 18, 2], [3, 6, 12, 24]),
    'base': (128, [2, 2, 18, 2], [4, 8, 16, 32]),
    'large': (192, [2, 2, 18, 2], [6, 12, 24, 48]),
}


def _backbone_and_decoder(pretrained, args, allow_lazy):
    swin_type = getattr(args, 'swin_type', 'base')
    if swin_type not in SWIN_VARIANTS:
        raise AssertionError(f"unknown swin_type {swin_type!r}")
    embed_dim, depths, num_heads = SWIN_VARIANTS[swin_type]
    window_size = 12 if ('window12' in pretrained or getattr(args, 'window12', False)) else 7
    mha_s = getattr(args, 'mha', '')
    mha = [int(a) for a in mha_s.split('-')] if mha_s else [1, 1, 1, 1]
    out_indices = (1, 2, 3) if (allow_lazy and getattr(args, 'lazy_pred', False)) else (0, 1, 2, 3)
    backbone = MultiModalSwinTransformer(embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                                         ape=False, drop_path_rate=getattr(args, 'drop_path_rate', 0.3), patch_norm=True,
                                         out_indices=out_indices, use_checkpoint=False, num_heads_fusion=mha,
                                         fusion_drop=getattr(args, 'fusion_drop', 0.0), args=args)
    backbone.init_weights(pretrained=pretrained if pretrained else None)
    return backbone, SimpleDecoding(8 * embed_dim, args)


def lavt(pretrained='', args=None):
    backbone, classifier = _backbone_and_decoder(pretrained, args, allow_lazy=False)
    return LAVT(backbone, classifier)


def lavt_one(pretrained='', args=None):
    backbone, classifier = _backbone_and_decoder(pretrained, args, allow_lazy=True)
    return LAVTOne(backbone, classifier, args)


VIDEO_SWIN_VARIANTS = {    # lib/segmentation.py:156-172 (embed_dim, depths, heads, drop_path_rate)
    'tiny': (96, [2, 2, 6, 2], [3, 6, 12, 24], 0.1),
    'small': (96, [2, 2, 18, 2], [3, 6, 12, 24], 0.2),
    'base': (128, [2, 2, 18, 2], [4, 8, 16, 32], 0.3),
}


def lavt_video(pretrained='', args=None):
    """Reference lib/segmentation.py:153-221: Video-Swin (patch (1,4,4), window (8,7,7) or (8,12,12)) + SimpleDecoding + LAVTVideo."""
    swin_type = getattr(args, 'swin_type', 'base')
    if swin_type not in VIDEO_SWIN_VARIANTS:
        raise AssertionError(f"unknown swin_type {swin_type!r}")
    embed_dim, depths, num_heads, drop_path_rate = VIDEO_SWIN_VARIANTS[swin_type]
    window_size = (8, 12, 12) if getattr(args, 'window12', False) else (8, 7, 7)
    mha_s = getattr(args, 'mha', '')
    mha = [int(a) for a in mha_s.split('-')] if mha_s else [1, 1, 1, 1]
    out_indices = (1, 2, 3) if getattr(args, 'lazy_pred', False) else (0, 1, 2, 3)
    backbone = MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=embed_dim, depths=depths, num_heads=num_heads,
                                           window_size=window_size, drop_path_rate=drop_path_rate, patch_norm=True,
                                           out_indices=out_indices, use_checkpoint=getattr(args, 'use_checkpoint', False),
                                           num_heads_fusion=mha, fusion_drop=getattr(args, 'fusion_drop', 0.0), args=args)
    backbone.init_weights(pretrained=pretrained if pretrained else None)
    return LAVTVideo(backbone, SimpleDecoding(8 * embed_dim, args), args)
