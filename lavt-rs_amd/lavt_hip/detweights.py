"""Deterministic, name-keyed weight generator.

Fixtures and benchmarks never store model weights (184-475 MB); instead every
tensor of a state dict is regenerated from its *name* and *shape*:
seed = crc32(name) ^ salt, values from numpy PCG64.  The scales are chosen so
activations stay O(1) through the whole network and the two output logits are
well separated (|logit1-logit0| >> 1e-3 on almost every pixel), which makes the
"identical argmax mask" parity gate meaningful.

The same generator feeds: tests/golden/make_golden.py (reference side),
oracle/, the HIP model (tests, smoke) and bench.py.
"""
import math
import zlib

import numpy as np
import torch


def _rng(name: str, salt: int):
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1)) & 0xFFFFFFFF))


def det_tensor(name: str, shape, dtype=torch.float32, salt: int = 0) -> torch.Tensor:
    """Return the deterministic value for state-dict entry `name`."""
    shape = tuple(int(s) for s in shape)
    g = _rng(name, salt)
    leaf = name.rsplit(".", 1)[-1]

    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "relative_position_index":
        raise KeyError("index buffers are structural, not generated")

    def normal(std):
        return torch.from_numpy(g.standard_normal(shape).astype(np.float32) * np.float32(std))

    is_norm = (".norm" in name or name.startswith("norm") or ".bn" in name or "patch_embed.norm" in name or "LayerNorm" in name)
    if leaf == "running_mean":
        t = normal(0.1)
    elif leaf == "running_var":
        t = 1.0 + 0.3 * normal(1.0).abs()
    elif leaf == "relative_position_bias_table":
        t = normal(0.5)
    elif is_norm and leaf == "weight":
        t = 1.0 + normal(0.1)
    elif is_norm and leaf == "bias":
        t = normal(0.1)
    elif leaf == "bias":
        t = normal(0.1)
        if name.endswith("classifier.conv1_1.bias"):
            t = torch.tensor([0.35, -0.35], dtype=torch.float32)[: shape[0]]
    elif leaf == "weight" or leaf.endswith("_embed"):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        std = 1.0 / math.sqrt(max(fan_in, 1))
        if "res_gate" in name:
            std *= 1.5
        if name.endswith("classifier.conv1_1.weight"):
            std *= 4.0
        t = normal(std)
    else:
        t = normal(0.05)
    return t.to(dtype)


def fill_state_dict_(module: torch.nn.Module, salt: int = 0) -> None:
    """Overwrite every parameter / float buffer of `module` in place (index buffers untouched)."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if name.endswith("relative_position_index"):
                continue
            t.copy_(det_tensor(name, t.shape, t.dtype, salt).to(t.device))


def det_inputs(batch: int, size: int, n_l: int = 20, seed: int = 1234, frames: int = 0):
    """Synthetic batch of SURVEY.md 8d: image randn, language randn(B,768,n_l), ragged l_mask, target."""
    g = torch.Generator("cpu").manual_seed(seed)
    if frames:
        x = torch.randn(batch, frames, 3, size, size, generator=g)
    else:
        x = torch.randn(batch, 3, size, size, generator=g)
    l = torch.randn(batch, 768, n_l, generator=g)
    lens = torch.randint(5, n_l + 1, (batch,), generator=g)
    l_mask = (torch.arange(n_l)[None, :] < lens[:, None]).float().unsqueeze(-1)
    nb = batch * max(frames, 1)
    target = torch.randint(0, 2, (nb, size, size), generator=g)
    return x, l, l_mask, target
