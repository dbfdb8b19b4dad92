"""Deterministic fake 'public Swin' checkpoints for the checkpoint-surgery tests (shared by tests/golden/make_golden.py, which feeds
them to the reference's loaders, and tests/test_host_logic.py, which feeds them to lavt_hip.checkpoint)."""
import torch

from lavt_hip.detweights import det_tensor


def synthetic_swin_checkpoint(embed=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), ws=5, prefix="module.backbone.", patch_t=None, index_n=None):
    """A fake 'public Swin' checkpoint (window `ws`) with deterministic values: only the tensors whose handling the loaders differ on."""
    sd = {}
    C = embed
    for i, (d, h) in enumerate(zip(depths, heads)):
        for b in range(d):
            k = f"layers.{i}.blocks.{b}.attn."
            sd[prefix + k + "relative_position_bias_table"] = det_tensor("ckpt." + k + "table", ((2 * ws - 1) ** 2, h), torch.float32)
            n_idx = index_n or ws * ws
            sd[prefix + k + "relative_position_index"] = torch.zeros(n_idx, n_idx, dtype=torch.long)
            sd[prefix + k + "qkv.weight"] = det_tensor("ckpt." + k + "qkv.weight", (3 * C, C), torch.float32)
        C *= 2
    shape = (embed, 3, 4, 4) if patch_t is None else (embed, 3, patch_t, 4, 4)
    sd[prefix + "patch_embed.proj.weight"] = det_tensor("ckpt.patch_embed.proj.weight", shape, torch.float32)
    sd[prefix + "norm.weight"] = det_tensor("ckpt.norm.weight", (C // 2,), torch.float32)        # the 2-D checkpoints' final norm: unexpected key
    if prefix:
        sd["module.decode_head.conv_seg.weight"] = torch.zeros(2, 4)                              # upper-net tensor: must be dropped
    return sd


def synthetic_lavt2d_checkpoint():
    """A fake 'released 2-D LAVT' state dict (micro dims, window 5; `backbone.` / `classifier.` prefixes, index buffers included): the tensors
    of the reference's 2-D LAVT micro model (key list written by make_golden.py) with name-keyed deterministic values."""
    import os
    sd = {}
    for line in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_keys_lavt2d_micro_w5.txt")):
        k, shp, kind = line.strip().split("|")
        shape = tuple(int(s) for s in shp.split("x")) if shp else ()
        sd[k] = torch.zeros(shape, dtype=torch.long) if kind == "i" else det_tensor("lavt2d." + k, shape, torch.float32)
    return sd
