"""Pin the CPU oracle (oracle/lavt_oracle.py) to golden vectors captured from the real reference
(tests/golden/make_golden.py).  fp32 CPU vs fp32 CPU: tolerance 2e-5 abs on O(1) activations."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lavt_hip.detweights import det_inputs, det_tensor
from oracle import lavt_oracle as O

TOL = 2e-5


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator("cpu").manual_seed(seed))


def gen_sd(spec, prefix=""):
    """spec: {key: shape} -> deterministic tensors keyed by the *unprefixed* name the reference module used."""
    return {prefix + k: det_tensor(k, s) for k, s in spec.items()}


def attn_spec(C, nH, ws, p=""):
    return {p + "relative_position_bias_table": ((2 * ws - 1) ** 2, nH), p + "qkv.weight": (3 * C, C), p + "qkv.bias": (3 * C,),
            p + "proj.weight": (C, C), p + "proj.bias": (C,)}


def block_spec(C, nH, ws):
    s = {"norm1.weight": (C,), "norm1.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,),
         "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,), "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,)}
    s.update(attn_spec(C, nH, ws, "attn."))
    return s


def pwam_spec(C, p=""):
    s = {}
    for k, cin in (("vis_project.0", C), ("project_mm.0", C), ("image_lang_att.f_query.0", C), ("image_lang_att.W.0", C),
                   ("image_lang_att.f_key.0", 768), ("image_lang_att.f_value.0", 768)):
        s[p + k + ".weight"] = (C, cin, 1)
        s[p + k + ".bias"] = (C,)
    return s


def close(a, b, tol=TOL):
    b = torch.as_tensor(np.asarray(b))
    err = (a - b).abs().max().item()
    assert a.shape == b.shape and err <= tol, f"max abs err {err:.3e} > {tol}"


@pytest.mark.parametrize("tag", ["w7", "w12"])
def test_window_attention(golden, tag):
    g = golden(f"win_attn_{tag}")
    C, nH, ws, Bw, Hp = (int(g[k]) for k in ("C", "nH", "ws", "Bw", "Hp"))
    sd = gen_sd(attn_spec(C, nH, ws), "")
    sd = {"a." + k: v for k, v in sd.items()}
    x = randn(int(g["seed"]), Bw, ws * ws, C)
    close(O.window_attention(sd, "a", x, nH, ws, None), g["y_nomask"])
    close(O.window_attention(sd, "a", x, nH, ws, O.shift_mask(Hp, Hp, ws, ws // 2)), g["y_mask"])


def test_shift_masks(golden):
    g = golden("shift_masks")
    for Hp, ws in ((126, 7), (36, 12), (24, 12), (7, 7), (14, 7)):
        m = O.shift_mask(Hp, Hp, ws, ws // 2)
        assert list(m.shape) == g[f"n_{Hp}_{ws}"].tolist()
        assert np.array_equal(np.packbits((m != 0).numpy().reshape(-1)), g[f"m_{Hp}_{ws}"])
        assert set(m.unique().tolist()) <= {0.0, -100.0}


@pytest.mark.parametrize("tag", ["15_w12", "28_w7", "10_w7"])
@pytest.mark.parametrize("shifted", [0, 1])
def test_swin_block(golden, tag, shifted):
    g = golden(f"block_{tag}_s{shifted}")
    C, nH, ws, H, B = (int(g[k]) for k in ("C", "nH", "ws", "H", "B"))
    sd = {"b." + k: v for k, v in gen_sd(block_spec(C, nH, ws)).items()}
    x = randn(int(g["seed"]), B, H * H, C)
    close(O.swin_block(sd, "b", x, H, H, nH, ws, bool(shifted)), g["y"], 5e-5)


@pytest.mark.parametrize("tag", ["even", "odd"])
def test_patch_merging(golden, tag):
    g = golden(f"patch_merging_{tag}")
    H, W, C, B = (int(g[k]) for k in ("H", "W", "C", "B"))
    sd = {"d." + k: v for k, v in gen_sd({"reduction.weight": (2 * C, 4 * C), "norm.weight": (4 * C,), "norm.bias": (4 * C,)}).items()}
    close(O.patch_merging(sd, "d", randn(int(g["seed"]), B, H * W, C), H, W), g["y"])


def test_patch_embed(golden):
    g = golden("patch_embed")
    C = int(g["C"])
    sd = {"pe." + k: v for k, v in gen_sd({"proj.weight": (C, 3, 4, 4), "proj.bias": (C,), "norm.weight": (C,), "norm.bias": (C,)}).items()}
    x = randn(int(g["seed"]), *g["shape"].tolist())
    t, H4, W4 = O.patch_embed(sd, "pe", x)
    close(t.transpose(1, 2).reshape(x.shape[0], C, H4, W4), g["y"])


@pytest.mark.parametrize("G", [1, 2])
def test_pwam(golden, G):
    g = golden(f"pwam_g{G}")
    C, T = int(g["C"]), int(g["T"])
    sd = {"f." + k: v for k, v in gen_sd(pwam_spec(C)).items()}
    x, l = randn(int(g["seeds"][0]), 2, T, C), randn(int(g["seeds"][1]), 2, 768, 20)
    m = torch.zeros(2, 20, 1)
    for b, n in enumerate(g["valid"].tolist()):
        m[b, :n] = 1
    close(O.sila(sd, "f.image_lang_att", x, l, m, G), g["lang"], 5e-5)
    close(O.pwam(sd, "f", x, l, m, G), g["y"], 5e-5)


def test_stage(golden):
    g = golden("stage_10x9")
    C, nH, ws = 64, 2, 7
    spec = {}
    for b in range(2):
        spec.update({f"blocks.{b}.{k}": v for k, v in block_spec(C, nH, ws).items()})
    spec.update(pwam_spec(C, "fusion."))
    spec.update({"res_gate.0.weight": (C, C), "res_gate.2.weight": (C, C),
                 "downsample.reduction.weight": (2 * C, 4 * C), "downsample.norm.weight": (4 * C,), "downsample.norm.bias": (4 * C,)})
    sd = {"s." + k: v for k, v in gen_sd(spec).items()}
    x, l = randn(int(g["seeds"][0]), 2, 90, C), randn(int(g["seeds"][1]), 2, 768, 20)
    m = torch.zeros(2, 20, 1)
    for b, n in enumerate(g["valid"].tolist()):
        m[b, :n] = 1
    r, xd, H, W = O.stage(sd, "s", x, 10, 9, l, m, 2, nH, ws, last=False)
    assert [10, 9, H, W] == g["hw"].tolist()
    close(r, g["r"], 1e-4)
    close(xd, g["x_down"], 1e-4)


def dec_spec(c4):
    h = c4 // 2
    s = {}
    for lvl, cin in ((4, c4 + c4 // 2), (3, h + c4 // 4), (2, h + c4 // 8)):
        s[f"conv1_{lvl}.weight"] = (h, cin, 3, 3)
        s[f"conv2_{lvl}.weight"] = (h, h, 3, 3)
        for j in (1, 2):
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                s[f"bn{j}_{lvl}.{leaf}"] = (h,)
    s["conv1_1.weight"] = (2, h, 1, 1)
    s["conv1_1.bias"] = (2,)
    return s


def test_decoder(golden):
    g = golden("decoder_c64")
    sd = {"c." + k: v for k, v in gen_sd(dec_spec(64)).items()}
    feats = [randn(int(s), 2, c, hw, hw) for s, (c, hw) in zip(g["seeds"], ((64, 4), (32, 8), (16, 16), (8, 32)))]
    close(O.decoder(sd, "c", *feats, training=False), g["y_eval"], 5e-5)
    close(O.decoder(sd, "c", *feats, training=True), g["y_train"], 1e-4)


def full_sd(keys_file):
    import os
    from conftest import GOLDEN
    sd = {}
    for line in open(os.path.join(GOLDEN, keys_file)):
        k, shp = line.strip().split("|")
        if k.endswith("relative_position_index"):
            continue
        shape = tuple(int(s) for s in shp.split("x")) if shp else ()
        sd[k] = det_tensor(k, shape, torch.long if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def test_e2e_swin_t_224(golden):
    g = golden("e2e_swin_t_224")
    sd = full_sd("state_dict_keys_swin_t.txt")
    x, l, _, tgt = det_inputs(1, 224, 20, seed=int(g["seed"]))
    m = torch.zeros(1, 20, 1)
    m[0, : int(g["valid"])] = 1
    with torch.no_grad():
        feats = O.backbone(sd, "backbone", x, l, m, "tiny", 7)
        logits = O.lavt_forward(sd, x, l, m, "tiny", 7)
    close(feats[0][:, :, ::4, ::4], g["c1"], 2e-4)
    close(feats[1][:, :, ::2, ::2], g["c2"], 2e-4)
    close(feats[2], g["c3"], 2e-4)
    close(feats[3], g["c4"], 2e-4)
    close(logits, g["logits"], 2e-4)
    ref_logits = torch.as_tensor(g["logits"])
    decisive = (ref_logits[:, 1] - ref_logits[:, 0]).abs() > 2e-3
    ref_mask = torch.as_tensor(np.unpackbits(g["mask"])[: 224 * 224].reshape(1, 224, 224)).bool()
    assert torch.equal(logits.argmax(1).bool()[decisive], ref_mask[decisive])
    I, U = O.iou_counts(logits, tgt)
    assert abs(I - int(g["I"])) <= (~decisive).sum() and abs(U - int(g["U"])) <= (~decisive).sum()
    assert abs(float(O.weighted_ce(logits, tgt)) - float(g["loss"])) < 1e-5


def test_e2e_swin_b_w12(golden):
    g = golden("e2e_swin_b_w12_96")
    sd = full_sd("state_dict_keys_swin_b_w12.txt")
    x, l, m, tgt = det_inputs(2, 96, 20, seed=int(g["seed"]))
    with torch.no_grad():
        logits = O.lavt_forward(sd, x, l, m, "base", 12)
    close(logits, g["logits"], 3e-4)


def grad_digest(t, n=24):
    import torch.nn.functional as F
    f = t.detach().reshape(-1).double()
    step = max(f.numel() // (n // 2), 1)
    samp = torch.cat([f[: n // 2], f[::step][: n // 2]])
    samp = F.pad(samp, (0, n - samp.numel()))
    return torch.cat([torch.stack([f.norm(), f.sum()]), samp]).float()


def micro_sd():
    """State dict of the test-only 'micro' LAVT (embed 32, depths 2-2-2-2, heads 1-2-4-8, window 7)."""
    sd = {}
    C0 = 32
    sd.update({"backbone.patch_embed." + k: s for k, s in {"proj.weight": (C0, 3, 4, 4), "proj.bias": (C0,), "norm.weight": (C0,), "norm.bias": (C0,)}.items()})
    for i, nH in enumerate((1, 2, 4, 8)):
        C = C0 * 2 ** i
        p = f"backbone.layers.{i}."
        for b in range(2):
            sd.update({f"{p}blocks.{b}.{k}": s for k, s in block_spec(C, nH, 7).items()})
        sd.update(pwam_spec(C, p + "fusion."))
        sd[p + "res_gate.0.weight"] = (C, C)
        sd[p + "res_gate.2.weight"] = (C, C)
        if i < 3:
            sd.update({p + "downsample.reduction.weight": (2 * C, 4 * C), p + "downsample.norm.weight": (4 * C,), p + "downsample.norm.bias": (4 * C,)})
        sd[f"backbone.norm{i}.weight"] = (C,)
        sd[f"backbone.norm{i}.bias"] = (C,)
    sd.update({"classifier." + k: s for k, s in dec_spec(8 * C0).items()})
    return {k: det_tensor(k, s) for k, s in sd.items()}


def test_e2e_micro_train_grads(golden):
    """Forward + weighted CE + backward of the micro model in train mode (BN batch stats): every
    parameter gradient of the oracle's autograd against digests captured from the reference."""
    g = golden("e2e_tiny_train")
    sd = micro_sd()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "running_" not in k}
    full = dict(sd)
    full.update(params)
    x, l, m, tgt = det_inputs(2, 64, 20, seed=int(g["seed"]))
    x.requires_grad_(True)
    l.requires_grad_(True)
    logits = O.lavt_forward(full, x, l, m, "micro", 7, training=True)
    close(logits.detach(), g["logits"], 2e-4)
    loss = O.weighted_ce(logits, tgt)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    nograd = set(g["nograd"].tolist())
    assert nograd == {"backbone.layers.3.res_gate.0.weight", "backbone.layers.3.res_gate.2.weight"}
    checked = 0
    for k, p in params.items():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        ref = torch.as_tensor(g["g|" + k])
        got = grad_digest(p.grad)
        scale = max(float(ref[0]), 1e-6)          # l2 norm of the reference gradient
        assert float((got - ref).abs().max()) <= 2e-4 * scale + 2e-6, k   # f_value.bias grads are analytically 0 (IN removes them)
        checked += 1
    assert checked == len(params) - 2
    for name, t in (("dx", x.grad), ("dl", l.grad)):
        ref = torch.as_tensor(g[name])
        assert float((grad_digest(t) - ref).abs().max()) <= 2e-4 * float(ref[0]) + 1e-7


# ================================================================================================ video path
from oracle import lavt_video_oracle as OV  # noqa: E402


def sd_from_keys(keys_file):
    import os
    from conftest import GOLDEN
    sd = {}
    for line in open(os.path.join(GOLDEN, keys_file)):
        k, shp = line.strip().split("|")
        if k.endswith("relative_position_index"):
            continue
        shape = tuple(int(s) for s in shp.split("x")) if shp else ()
        sd[k] = det_tensor(k, shape, torch.long if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def test_video_masks(golden):
    g = golden("video_masks")
    for tag in "abcd":
        Dp, Hp, Wp, wd, wh, ww, sd_, sh, sw = g["cfg_" + tag].tolist()
        m = OV.shift_mask_3d(Dp, Hp, Wp, (wd, wh, ww), (sd_, sh, sw))
        assert list(m.shape) == g["n_" + tag].tolist()
        assert np.array_equal(np.packbits((m != 0).numpy().reshape(-1)), g["m_" + tag])


def vblock_spec(C, nH, window):
    s = block_spec(C, nH, 7)
    s["attn.relative_position_bias_table"] = ((2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1), nH)
    return s


@pytest.mark.parametrize("tag", ["t8", "t3", "t16"])
@pytest.mark.parametrize("shifted", [0, 1])
def test_video_block(golden, tag, shifted):
    g = golden(f"vblock_{tag}_s{shifted}")
    B, D, H, W = g["dims"].tolist()
    sd = {"b." + k: v for k, v in gen_sd(vblock_spec(64, 2, (8, 7, 7))).items()}
    x = randn(int(g["seed"]), B, D, H, W, 64)
    close(OV.swin_block_3d(sd, "b", x, 2, (8, 7, 7), bool(shifted)), g["y"], 1e-4)


@pytest.mark.parametrize("shifted", [0, 1])
def test_video_block_window12(golden, shifted):
    """`--window12` video windows, (8, 12, 12) = 1152 tokens (lib/video_swin_transformer.py:137-168, SURVEY.md 8 shape row 4'), un-shifted / shifted"""
    g = golden(f"vblock_w12_s{shifted}")
    B, D, H, W = g["dims"].tolist()
    sd = {"b." + k: v for k, v in gen_sd(vblock_spec(64, 2, (8, 12, 12))).items()}
    x = randn(int(g["seed"]), B, D, H, W, 64)
    y = OV.swin_block_3d(sd, "b", x, 2, (8, 12, 12), bool(shifted))
    close(y[:, :, ::2, ::2], g["y"], 1e-4)
    assert abs(float(y.double().sum()) - float(g["ysum"])) <= 1e-5 * float(g["yabs"])


def sept_spec(C, p=""):
    s = {}
    for k, ks in (("temporal_vis_project.0", 3), ("spatial_vis_project.0", 1), ("f_query_t.0", 3), ("f_query_s.0", 1), ("W_t.0", 3), ("W_s.0", 1),
                  ("project_mm_t.0", 3), ("project_mm_s.0", 1)):
        s[p + k + ".weight"] = (C, C, ks, ks, ks)
        s[p + k + ".bias"] = (C,)
    for k in ("f_key.0", "f_value.0"):
        s[p + k + ".weight"] = (C, 768, 1)
        s[p + k + ".bias"] = (C,)
    return s


def test_video_sep_t_pwam(golden):
    g = golden("sep_t_pwam")
    sd = {"f." + k: v for k, v in gen_sd(sept_spec(32)).items()}
    x, l = randn(int(g["seeds"][0]), 2, 4, 6, 5, 32), randn(int(g["seeds"][1]), 2, 768, 20)
    m = torch.zeros(2, 20, 1)
    for b, n in enumerate(g["valid"].tolist()):
        m[b, :n] = 1
    close(OV.sep_t_pwam(sd, "f", x, l, m), g["y"], 1e-4)


@pytest.mark.parametrize("tag", ["pwam", "sept"])
def test_video_e2e_micro_train_grads(golden, tag):
    """The oracle runs in float64 here: the digests hold a plain sum of each gradient, and through 27-tap convolutions plus
    instance norms a float32 run carries ~1e-2 of cancellation noise in it (the reference's float32 run sits within 1e-3 of
    the float64 value).  Fixture batch is 2 -- see make_golden.py for the batch-1 PyTorch CPU instance-norm backward defect."""
    g = golden(f"e2e_video_micro_{tag}")
    sd = sd_from_keys(f"state_dict_keys_video_micro_{tag}.txt")
    dt = torch.float64
    params = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    full = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    full.update(params)
    frames, l, m, tgt = det_inputs(2, 64, 22, seed=int(g["seed"]), frames=4)
    frames, l, m = frames.to(dt).requires_grad_(True), l.to(dt).requires_grad_(True), m.to(dt)
    logits = OV.lavt_video_forward(full, frames, l, m, "micro", (8, 7, 7), sep_t=(tag == "sept"), training=True)
    close(logits.detach().float(), g["logits"], 3e-4)
    loss = F.cross_entropy(logits, tgt, weight=torch.tensor([0.9, 1.1], dtype=dt))
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    nograd = set(g["nograd"].tolist())
    tol = 2e-3
    for k, p in params.items():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        ref = torch.as_tensor(g["g|" + k])
        assert float((grad_digest(p.grad.float()) - ref).abs().max()) <= tol * max(float(ref[0]), 1e-6) + 3e-6, k
    for name, t in (("dframes", frames.grad), ("dl", l.grad)):
        ref = torch.as_tensor(g[name])
        assert float((grad_digest(t.float()) - ref).abs().max()) <= tol * float(ref[0]) + 3e-6


# ---------------------------------------------------------------------------------------------- text side
from oracle import bert_oracle as OB  # noqa: E402


@pytest.mark.parametrize("N", [20, 22])
def test_bert_micro(golden, N):
    """oracle/bert_oracle.py against vectors of the installed `transformers.BertModel` (the reference's `./bert` copy of transformers 3.0.2
    is absent): last_hidden_state and gradient digests on name-keyed deterministic weights."""
    g = golden(f"bert_micro_n{N}")
    sd = sd_from_keys("state_dict_keys_bert_micro.txt")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ids, mask = torch.as_tensor(g["ids"]), torch.as_tensor(g["mask"])
    out = OB.bert_forward(params, ids, mask, num_heads=2)
    close(out.detach(), g["out"], 2e-5)
    (out * torch.as_tensor(g["w"])).sum().backward()
    for key in g.files:
        if not key.startswith("grad__"):
            continue
        k = key[len("grad__"):].replace("__", ".")
        ref = torch.as_tensor(g[key])
        assert float((grad_digest(params[k].grad) - ref).abs().max()) <= 1e-4 * max(float(ref[0]), 1e-6) + 3e-6, k


def test_pad_ids():
    ids, mask = OB.pad_ids([101, 7, 8, 9, 102], 8)
    assert ids == [101, 7, 8, 9, 102, 0, 0, 0] and mask == [1, 1, 1, 1, 1, 0, 0, 0]
    ids, mask = OB.pad_ids(range(1, 30), 20)
    assert ids == list(range(1, 21)) and mask == [1] * 20


# ---------------------------------------------------------------------------------------------- full-size configurations (BASELINE.json configs 2-4)
def _full_oracle_check(g, lowres, logits, tgt, tol=2e-4):
    close(lowres.detach(), g["lowres"], tol)
    stride = 4 if logits.shape[0] <= 2 else 8
    close(logits.detach()[:, :, 1::stride, 2::stride], g["logits_s"], tol)
    n = logits.shape[0] * logits.shape[-1] * logits.shape[-2]
    ref_mask = torch.as_tensor(np.unpackbits(g["mask"])[:n].reshape(logits.shape[0], *logits.shape[-2:])).bool()
    decisive = torch.as_tensor(np.unpackbits(g["decisive"])[:n].reshape(ref_mask.shape)).bool()
    pred = logits.argmax(1).bool()
    assert torch.equal(pred[decisive], ref_mask[decisive])
    assert abs(float(O.weighted_ce(logits, tgt)) - float(g["loss"])) < 1e-5


@pytest.mark.parametrize("tag,keys,variant,ws,B", [("full_swin_b_480_b2", "state_dict_keys_swin_b_w12.txt", "base", 12, 2),
                                                     ("full_swin_t_480_b8", "state_dict_keys_swin_t.txt", "tiny", 7, 8)])
def test_full_size_image_forward(golden, tag, keys, variant, ws, B):
    """The oracle at the sizes bench.py times (Swin-B w12 2x480x480, Swin-T w7 8x480x480; train-mode BatchNorm) against the reference's
    own run of those configurations: forward only here (the GPU suite checks every gradient against the same fixtures)."""
    g = golden(tag)
    sd = sd_from_keys(keys)
    x, l, m, tgt = det_inputs(B, 480, 20, seed=int(g["seed"]))
    with torch.no_grad():
        c1, c2, c3, c4 = O.backbone(sd, "backbone", x, l, m, variant, ws)
        lowres = O.decoder(sd, "classifier", c4, c3, c2, c1, training=True)
        logits = F.interpolate(lowres, size=(480, 480), mode="bilinear", align_corners=True)
    _full_oracle_check(g, lowres, logits, tgt)


@pytest.mark.parametrize("tag", ["pwam", "sept"])
def test_full_size_video_forward(golden, tag):
    """Video-Swin-B, T=8 at 384x384 (BASELINE configs[3]), PWAM and the README SepTPWAM recipe: oracle forward vs the reference's"""
    from oracle import lavt_video_oracle as OV
    g = golden(f"full_video_{tag}_t8_384")
    sd = {"backbone." + k: v for k, v in sd_from_keys(f"state_dict_keys_video_swin_b_{tag}.txt").items()}
    sd = {k: det_tensor(k, v.shape, v.dtype) for k, v in sd.items()}                     # names carry the 'backbone.' prefix in the fixture's model
    sd.update({k: v for k, v in sd_from_keys("state_dict_keys_swin_b_w12.txt").items() if k.startswith("classifier.")})
    frames, l, m, tgt = det_inputs(1, 384, 22, seed=int(g["seed"]), frames=8)
    with torch.no_grad():
        c1, c2, c3, c4 = OV.backbone_3d(sd, "backbone", frames.permute(0, 2, 1, 3, 4), l, m, "base", (8, 7, 7), sep_t=(tag == "sept"))
        lowres = O.decoder(sd, "classifier", c4, c3, c2, c1, training=True)
        logits = F.interpolate(lowres, size=(384, 384), mode="bilinear", align_corners=True)
    # SepTPWAM (five 27-tap convolutions + instance norms per stage, 24 blocks): float32 rounding alone is worth a few 1e-4 on logits of
    # sigma 2.2 -- measured here: reference fp32 vs a float64 evaluation 2.8e-4, this oracle in fp32 vs float64 4.6e-4
    _full_oracle_check(g, lowres, logits, tgt, 1e-3 if tag == "sept" else 2e-4)


@pytest.mark.parametrize("tag", ["a", "b", "same"])
def test_multiclass_dice(golden, tag):
    """oracle.multiclass_dice vs the reference's losses.MultiClassDiceLoss (after the align_corners upsample): loss and gradient"""
    g = golden(f"dice_{tag}")
    B, h, w, H, W = g["dims"].tolist()
    y = (randn(int(g["seeds"][0]), B, 2, h, w) * 2.0).requires_grad_(True)
    tgt = (randn(int(g["seeds"][1]), B, H, W) > 0.3).long()
    tgt[B - 1] = 0
    loss = O.multiclass_dice(F.interpolate(y, size=(H, W), mode="bilinear", align_corners=True), tgt)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    close(y.grad, g["dy"], 1e-7)
