#!/usr/bin/env python3
"""LAVT hot-path benchmark on MI355X.   python bench.py --gpus N --steps K --warmup W
(N > 1: one rank per GPU over RCCL -- launched by torch.distributed.run, or, when started by hand without RANK in the environment,
bench.py starts its own N worker processes before touching a GPU.)

Workload (BASELINE.json metric: train images/s, 480x480 Swin-B LAVT): Swin-B window-12 LAVT, bf16 compute,
batch 2 per GPU (configs[2] of BASELINE.json, the per-GPU shard of the headline config; weak scaling), synthetic
480x480 images + 20-token language embeddings, deterministic random-init weights.  One step = forward +
weighted cross-entropy (fused with the final upsample: lavt_upsample_ce_*) + backward (+ gradient all-reduce when N > 1); the optimizer is excluded.

Prints ONE JSON line on rank 0 with the driver's contract plus
  "roofline":     the kernel family with the largest us/step of this step, timed per launch with HIP events inside eager steps (profile_step),
  "roofline_conv": the decoder conv2_2 implicit GEMM timed alone (round 1's roofline entry, kept for continuity),
  "cpu_baseline": the CPU oracle (oracle/lavt_oracle.py, a port of the reference) timed on the host cores (N=1, rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "lavt-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_IMAGE = {"swin_b_w12_480": 394.57, "swin_t_w7_480": 172.43,        # SURVEY.md 8 (2*MAC, padded tokens counted)
                       "video_swin_b_t8_384": 2071.5 / 8, "video_swin_b_t8_384_sept": 3115.2 / 8}   # per frame (G/clip / 8)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
BF16_DENSE_PEAK_TFLOPS = 2500.0                                                 # MI355X_MICROARCH.md (dense, no sparsity)

WORKLOADS = {
    "swin_b_w12_480_b2": dict(variant="base", window12=True, batch=2, size=480, flops="swin_b_w12_480"),
    "swin_t_w7_480_b8": dict(variant="tiny", window12=False, batch=8, size=480, flops="swin_t_w7_480"),
    # BASELINE.json configs[4]: fp8 (e4m3) weights / activations for the forward contractions that opt in, batch 4 per GPU (run with --dtype fp8)
    "swin_b_w12_480_b4_fp8": dict(variant="base", window12=True, batch=4, size=480, flops="swin_b_w12_480", fp8=True),
    "swin_b_w12_480_b4": dict(variant="base", window12=True, batch=4, size=480, flops="swin_b_w12_480"),
    # the same step with the BERT-base text encoder inside the model (`lavt_one`): token ids in, BERT trained with the rest
    "lavt_one_swin_b_w12_480_b2": dict(variant="base", window12=True, batch=2, size=480, flops="swin_b_w12_480", one=True),
    # BASELINE.json configs[3]: Video-Swin-B LAVT, one clip of T=8 frames at 384x384 per GPU (metric counts frames); PWAM / README SepTPWAM recipe
    "video_swin_b_t8_384": dict(variant="base", video=True, frames=8, batch=1, size=384, flops="video_swin_b_t8_384", sept=False),
    "video_swin_b_t8_384_sept": dict(variant="base", video=True, frames=8, batch=1, size=384, flops="video_swin_b_t8_384_sept", sept=True),
}


def build_model(cfg, device, drop_path=0.3):
    from types import SimpleNamespace
    from lavt_hip.detweights import fill_state_dict_
    from lib import segmentation
    if cfg.get("video"):
        flags = dict(sep_t_pwam=True, conv3d_kernel_size_t="3-3-3", conv3d_kernel_size_s="1-1-1", w_t3x3_s1x1=True, mm_t3x3_s1x1=True) if cfg["sept"] else {}
        from lib.mask_predictor import SimpleDecoding
        from lib.video_swin_transformer import MultiModalSwinTransformer3D
        a = SimpleNamespace(**flags)
        bb = MultiModalSwinTransformer3D(patch_size=(1, 4, 4), embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7),
                                         drop_path_rate=drop_path, patch_norm=True, out_indices=(0, 1, 2, 3), num_heads_fusion=[1, 1, 1, 1], args=a)
        model = _VideoStep(bb, SimpleDecoding(1024, a))         # lavt_video without the BERT encoder: language features are the input, as for `lavt`
    else:
        args = SimpleNamespace(swin_type=cfg["variant"], window12=cfg["window12"], drop_path_rate=drop_path, bert_random_init=True)     # synthetic data: no checkpoint
        model = segmentation.lavt_one("", args) if cfg.get("one") else segmentation.lavt("", args)
    fill_state_dict_(model)
    return model.to(device)


class _VideoStep(torch.nn.Module):
    """LAVTVideo.forward_backbone as a (clip, l_feats, l_mask) -> low-resolution logits module for the step harness"""

    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone, self.classifier = backbone, classifier

    def forward_lowres(self, x, l, m):
        f = self.backbone(x.permute(0, 2, 1, 3, 4), l, m)
        return self.classifier(f[3], f[2], f[1], f[0])

    def forward(self, x, l, m):
        from lib._utils import _upsample_logits
        return _upsample_logits(self.forward_lowres(x, l, m), x.shape[-2:])


def measure_conv_kernel(device, iters=20):
    """Decoder conv2_2 of Swin-B at batch 2 alone: implicit GEMM M=2*120*120, N=512, K=9*512, bf16 MFMA.  Algorithmic flops
    per launch = 2*M*N*K; duration = HIP events around `iters` back-to-back launches on the launch stream."""
    from lavt_hip import ops
    B, H, W, Cin, Cout = 2, 120, 120, 512, 512
    x = torch.randn(B * H * W, Cin, device=device).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=device) * (9 * Cin) ** -0.5
    import lavt_hip
    f8 = lavt_hip.fp8_enabled()          # fp8 workload: the same convolution on e4m3 operands (v_mfma 16x16x128 f8f6f4: dense peak 5 PFLOP/s)
    with torch.no_grad():
        y_ref = ops.conv3x3(x, None, w, B, H, W)            # the product op (packs the weight copy; fp8: quantises x and the weight)
        # ... then the very launch it issues (ops._ConvTaps.forward), with the output buffer and the packed weight prepared once: per call the host spends
        # ~20 us on the parameter struct, so the 20 launches queue up and the events bracket GPU time (through the autograd Function the host needed
        # longer per call than the kernel runs)
        Wp = ops.weights.get(w, torch.bfloat16, "conv3")
        y = torch.empty_like(y_ref)
        launch = lambda: ops.gemm_nt(torch.bfloat16, B * H * W, Cout, 9 * Cin, x, Cin, Wp, 9 * Cin, y, Cout, conv=(H, W, Cin, 0, 1, 1, 3, 3))
        if f8:          # the e4m3 contraction the product op issued, operands quantised once (same |max| slot, not advanced in between: identical bytes)
            Wq, wa = ops.weights.get_fp8(w, "conv3")
            xq, ap = ops.fp8.quantize(x, id(w))
            launch = lambda: ops.gemm_nt(torch.uint8, B * H * W, Cout, 9 * Cin, xq, Cin, Wq, 9 * Cin, y, Cout, conv=(H, W, Cin, 0, 1, 1, 3, 3), deq=(ap, wa.data_ptr()))
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), "the timed launch is the product op's launch"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
    del y, y_ref
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * H * W * Cout * 9 * Cin
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, tsrc = None, None          # fabric-side bytes per launch from the committed rocprofv3 --pmc passes of this kernel: a recorded constant, not measured by this run
    try:
        for fn in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r02_pmc_conv_and_grouped_wgrad.json"):          # the newest committed pass of this kernel (tools/pmc_passes.sh conv_one)
            if os.path.exists(os.path.join(ROOT, "profiles", fn)):
                traffic = int(json.load(open(os.path.join(ROOT, "profiles", fn)))["conv_one"]["derived"]["fabric_bytes (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)"])
                tsrc = f"profiles/{fn} (recorded)"
                break
    except Exception:  # noqa: BLE001
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")))["traffic_bytes_per_launch"]
            tsrc = "profiles/pmc_dominant_kernel.json (recorded)"
        except Exception:  # noqa: BLE001
            pass
    peak = 2.0 * BF16_DENSE_PEAK_TFLOPS if f8 else BF16_DENSE_PEAK_TFLOPS
    if f8:
        traffic, tsrc = None, None
        try:
            traffic = int(json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_conv_fp8.json")))["conv_fp8_one"]["derived"]["fabric_bytes (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)"])
            tsrc = "profiles/r04_pmc_conv_fp8.json (recorded)"
        except Exception:  # noqa: BLE001
            pass
    return {"bound": "mfma", "kernel": "implicit-GEMM conv3x3 512->512 @120x120, batch 2 (decoder conv2_2), timed alone" + (" [fp8: the e4m3 contraction; in the step its activation operand is the twin the producing BatchNorm + ReLU / bilinear kernel wrote]" if f8 else ""),
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "avg_launch_us": round(ms * 1e3, 2), "flops_per_launch": flops, "traffic": traffic, "traffic_source": tsrc}


def measure_attainable_peak(device, n=8192, iters=10):
    """the bf16 rate this box attains on a chip-filling GEMM (n^3, random operands, HIP events around `iters` back-to-back launches),
    next to the nominal 2.5 PFLOP/s every `frac` in this line is quoted against: the vendor library through torch.matmul (hipBLASLt / rocBLAS -- a
    yardstick, never on the product path) and this library's own NT kernel.  tools/gemm_yardstick.py has the per-shape table (profiles/)."""
    from lavt_hip import ops
    bf = torch.bfloat16
    A, B = torch.randn(n, n, device=device).to(bf), torch.randn(n, n, device=device).to(bf)
    C = torch.empty(n, n, device=device, dtype=bf)
    res = {"unit": "TFLOP/s", "shape": f"{n}x{n}x{n} bf16, fp32 accumulate"}
    for key, fn in (("vendor_hipblaslt", lambda: torch.matmul(A, B.t(), out=C)), ("own_gemm_nt", lambda: ops.gemm_nt(bf, n, n, n, A, n, B, n, C, n))):
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[key] = round(2.0 * n ** 3 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e12, 1)
        except Exception as e:  # noqa: BLE001
            res[key] = f"{type(e).__name__}: {e}"
    return res


def replay_trace_launch_us(kernel_substr, grid):
    """average duration of a kernel in the newest committed rocprofv3 trace of hipGraph replays (profiles/rNN_z_by_shape_graph_replay.txt, written by
    tools/trace_by_shape.py): a recorded figure, quoted next to the eager-timed one of this run"""
    import glob
    import re
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_z*_by_shape_graph_replay.txt")), reverse=True):          # (r05_zzz_... = the second session's table sorts in front of r05_z_...)
        for line in open(fn):
            if kernel_substr in line and (grid is None or f"grid {grid}x" in line):          # (tables are sorted by us/step: the first match is the heaviest launch of that kernel)
                m_ = re.search(r"x\s+([0-9.]+) us\s+grid", line)
                if m_:
                    return float(m_.group(1)), "profiles/" + os.path.basename(fn) + " (recorded)"
    return None, None


def profile_step(step, cfg, device, reps=3):
    """Where the step's time goes, measured in this process: `reps` eager steps with every C-ABI launch bracketed by HIP events on the launch
    stream (lavt_hip._capi.prof).  Launches are grouped into families (entry point + problem shape); the family with the largest us/step is
    the step's dominant kernel and becomes `roofline`.  Scope labels give the W-MSA + PWAM share (qkv + QK^T + PV + proj + PWAM +
    gate = 88.6 GFLOP forward per Swin-B image, x3 for training)."""
    from lavt_hip import _capi as K
    fams, scopes = {}, {}
    for _ in range(reps):
        K.prof.start()
        step._body()
        recs = K.prof.stop()
        for name, sc, note, us in recs:
            shape = note["shape"] if note else ""
            fl = note["flops"] if note else 0.0
            f = fams.setdefault((name, shape), [0, 0.0, 0.0, 0.0])
            f[0] += 1
            f[1] += us
            f[2] += fl
            f[3] += note.get("bytes", 0.0) if note else 0.0
            members = note.get("members") if note else None
            if members:                                   # grouped weight gradients: split the launch's time over its members' scopes by flops
                tot = sum(m[1] for m in members) or 1.0
                for msc, mfl in members:
                    s_ = scopes.setdefault(msc, [0.0, 0.0])
                    s_[0] += us * mfl / tot
                    s_[1] += mfl
            else:
                s_ = scopes.setdefault(sc, [0.0, 0.0])
                s_[0] += us
                s_[1] += fl
    dump = os.environ.get("LAVT_PROFILE_DUMP")
    if dump:          # every (entry point, shape, scope) family of the last eager step, for tools / profiles
        by = {}
        for name, sc, note, us in recs:
            k = (name, note["shape"] if note else "", sc)
            e = by.setdefault(k, [0, 0.0])
            e[0] += 1
            e[1] += us
        with open(dump, "w") as f:
            for (name, shape, sc), (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{us:9.1f} us  {n:4d} x {us / n:7.1f}  {sc:8s} {name} {shape}\n")
    total_us = sum(f[1] for f in fams.values()) / reps
    top = sorted(fams.items(), key=lambda kv: -kv[1][1])
    table = [{"entry": k[0], "shape": k[1], "launches_per_step": round(v[0] / reps, 1), "us_per_step": round(v[1] / reps, 1),
              "tflops": round(v[2] / v[1] * 1e-6, 1) if v[2] else None, "gbs": round(v[3] / v[1] * 1e-3, 1) if v[3] else None} for k, v in top[:8]]
    # the family with the largest us/step that carries an algorithmic-work annotation (flops -> MFMA roof, bytes -> HBM roof); the host wrappers
    # annotate every GEMM, the attention kernels, LayerNorm and the fused PWAM kernels, so this is the true top family unless an unannotated
    # normalisation kernel leads (then `unannotated_top` names it)
    (dname, dshape), dv = next(((k, v) for k, v in top if v[2] > 0 or v[3] > 0), top[0])
    if dv[2] > 0:
        ach = dv[2] / dv[1] * 1e-6
        roof = {"bound": "mfma", "kernel": f"{dname} [{dshape}] -- largest us/step of the step, timed per launch inside eager steps", "achieved": round(ach, 2),
                "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4), "avg_launch_us": round(dv[1] / dv[0], 2),
                "flops_per_launch": dv[2] / dv[0], "launches_per_step": round(dv[0] / reps, 1), "us_per_step": round(dv[1] / reps, 1), "traffic": None}
    else:
        ach = dv[3] / dv[1] * 1e-3          # bytes / us -> GB/s
        roof = {"bound": "hbm", "kernel": f"{dname} [{dshape}] -- largest us/step of the step, timed per launch inside eager steps", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(dv[1] / dv[0], 2),
                "bytes_per_launch": dv[3] / dv[0], "launches_per_step": round(dv[0] / reps, 1), "us_per_step": round(dv[1] / reps, 1), "traffic": None}
    if top[0][0] != (dname, dshape):
        roof["unannotated_top"] = {"entry": top[0][0][0], "us_per_step": round(top[0][1][1] / reps, 1)}
    if dname in ("lavt_gemm_tn_grouped", "lavt_gemm_tn_grouped_ln"):          # (_ln: the same launch carrying norm1's LayerNorm backward as rider workgroups)
        # fabric-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE on
        # tools/wgrad_group_one.py, the same five problems in token order): a recorded constant, not a measurement of this run
        try:
            fn = next(f for f in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc_grouped_wgrad_and_wmsa.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pm = json.load(open(os.path.join(ROOT, "profiles", fn)))["wgrad_group_one"]["derived"]
            roof["traffic"] = int(pm["fabric_bytes (FETCH_SIZE KB x2 gfx950 correction + WRITE_SIZE KB)"])
            roof["traffic_source"] = f"profiles/{fn} (recorded)"
            # bf16 operands read once (fc2, fc1, proj, the 792 padded rows of dqkv, qkv) + fp32 gradients written once
            us_, src_ = (None, None)
            if cfg.get("name") == "swin_b_w12_480_b2":          # stage-2 launch: 198 tiles + 225 rider workgroups on the pipelined kernel (round 5); 1242 blocks on the 64x64 launch before
                us_, src_ = replay_trace_launch_us("gemm_tn_pipe_kernel<4, 64>", None)
                if not us_:
                    us_, src_ = replay_trace_launch_us("gemm_tn_v2_grouped_ln_kernel<true, 2, 64, 1>", 1242)
            if us_:          # the rocprofv3 trace of graph replays is the figure the documents quote; the eager-timed one above is this run's own
                roof["avg_launch_us_graph_replay_trace"] = us_
                roof["frac_graph_replay_trace"] = round(roof["flops_per_launch"] / (us_ * 1e-6) / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4)
                roof["graph_replay_trace_source"] = src_
            roof["algorithmic_bytes"] = int(2 * (2 * 1800 * (512 + 2048) + 1800 * (512 + 512) + 792 * 1536 + 1800 * (1536 + 512)) + 4 * (2 * 2048 * 512 + 512 * 512 + 1536 * 512))
        except Exception:  # noqa: BLE001
            pass
    scope_us = {k: round(v[0] / reps, 1) for k, v in sorted(scopes.items(), key=lambda kv: -kv[1][0])}
    out = {"eager_kernel_us_per_step": round(total_us, 1), "top_families": table, "scope_us_per_step": scope_us}
    if cfg["flops"] == "swin_b_w12_480":
        wp_us = (scopes.get("wmsa", [0, 0])[0] + scopes.get("pwam", [0, 0])[0]) / reps
        wp_flops = 3.0 * 88.6e9 * cfg["batch"]
        out["wmsa_pwam_us_per_step"] = round(wp_us, 1)
        out["wmsa_pwam_mfma_frac"] = round(wp_flops / (wp_us * 1e-6) / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4) if wp_us else None
    return roof, out


def loss_check(model, step, cfg, device):
    """The captured step's arithmetic against the reference: one eager bf16 forward + fused loss with DropPath off (its random draws cannot be
    matched) on the rank-0 inputs, compared with the loss of the REFERENCE's fp32 CPU run of the same inputs (tests/golden/full_*.npz, captured by
    tests/golden/make_golden.py).  Gate: |d loss| <= 2e-2 (bf16 forward; the fp32 path meets 1e-4 in tests/)."""
    import numpy as np
    name = {"swin_b_w12_480_b2": "full_swin_b_480_b2", "swin_t_w7_480_b8": "full_swin_t_480_b8"}.get(cfg.get("name"))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz") if name else None
    if not path or not os.path.exists(path):
        return None
    ref = float(np.load(path)["loss"])
    from lib.backbone import DropPath
    from lib._utils import fused_loss
    dps = [m for m in model.modules() if isinstance(m, DropPath)]
    saved = [m.drop_prob for m in dps]
    bns = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    try:
        for m in dps:
            m.drop_prob = 0.0
        with torch.no_grad():
            loss, stats = fused_loss(model.forward_lowres(step.x, step.l, step.m), step.t, (0.9, 1.1))
        got = float(loss)
    finally:
        for m, p_ in zip(dps, saved):
            m.drop_prob = p_
        model.load_state_dict(bns, strict=False)
    return {"bf16_eager_drop_path_0": round(got, 5), "reference_fp32_cpu": round(ref, 5), "abs_diff": round(abs(got - ref), 5), "gate": 2e-2,
            "ok": bool(abs(got - ref) <= 2e-2)}


def cpu_baseline(cfg):
    """The CPU oracle (fp32 port of the reference path) on this host, as BASELINE.md 3 states it: the workload's batch (B=2 for Swin-B),
    forward + weighted CE + backward, 1 warm-up then the median of 3."""
    from lavt_hip.detweights import det_inputs, det_tensor
    from oracle import lavt_oracle as O
    keys = os.path.join(ROOT, "tests", "golden", "state_dict_keys_swin_b_w12.txt" if cfg["variant"] == "base" else "state_dict_keys_swin_t.txt")
    sd = {}
    for line in open(keys):
        k, shp = line.strip().split("|")
        if k.endswith("relative_position_index"):
            continue
        shape = tuple(int(s) for s in shp.split("x")) if shp else ()
        t = det_tensor(k, shape, torch.long if k.endswith("num_batches_tracked") else torch.float32)
        sd[k] = t.requires_grad_(True) if t.dtype.is_floating_point and "running_" not in k else t
    B = min(cfg["batch"], 2)
    x, l, m, tgt = det_inputs(B, cfg["size"], 20, seed=1234)
    ws = 12 if cfg["window12"] else 7
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        loss = O.weighted_ce(O.lavt_forward(sd, x, l, m, cfg["variant"], ws, training=True), tgt)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            if v.dtype.is_floating_point and v.grad is not None:
                v.grad = None
        if sum(times) > 90.0 and it >= 1:          # slow host: keep the default run within minutes
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"batch of {B} images {cfg['size']}x{cfg['size']}, fp32 forward+CE+backward of the CPU oracle (oracle/lavt_oracle.py): 1 warm-up + "
                      f"{len(timed)} timed, median {med:.2f} s (all: {', '.join(f'{t:.2f}' for t in times)})"}


def spawn_ranks(a):
    """`python bench.py --gpus N` started by hand (no torch.distributed.run): start N fresh worker processes, one per GPU, BEFORE anything in this
    process touches a GPU (no re-exec of a process that has initialised HIP); relay rank 0's JSON line.  The children are supervised: the first
    non-zero exit (or Ctrl-C, or the overall timeout) terminates the others -- a rank that dies at start-up must not leave its peers inside an RCCL
    rendezvous holding their GPUs -- and the parent exits non-zero."""
    import socket
    import subprocess
    import tempfile
    import time as _time
    lsock = socket.socket()
    lsock.bind(("127.0.0.1", 0))
    port = lsock.getsockname()[1]
    out0 = tempfile.TemporaryFile()
    procs = []
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       LOCAL_WORLD_SIZE=str(a.gpus), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            if r == 0:
                lsock.close()                # released only now: the window in which another process can take the port is the rank-0 start-up
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = _time.time() + float(os.environ.get("LAVT_BENCH_TIMEOUT_S", "1800"))
        bad = []
        while True:
            rcs = [p_.poll() for p_ in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc == 0 for rc in rcs):
                break
            if _time.time() > deadline:
                bad = [("timeout", -1)]
                break
            _time.sleep(0.2)
    except KeyboardInterrupt:
        bad = [("interrupted", -2)]
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(timeout=10)
            except Exception:  # noqa: BLE001
                p_.kill()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if bad:
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="swin_b_w12_480_b2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32", "fp8"], help="default: bf16 (fp8 for the *_fp8 workload)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-family kernel timing pass (roofline falls back to the conv timed alone)")
    ap.add_argument("--drop-path", type=float, default=0.3)
    ap.add_argument("--rccl-channels", type=int, default=0, help="N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS).  Every channel is a persistent "
                    "workgroup on a CU while a collective runs; the step's compute kernels are sized to about one round of the 256 CUs, so each CU RCCL "
                    "holds turns a 1.0-round launch into 2.0 rounds for the kernels it overlaps with: fewer channels = less overlap tax, lower ring bandwidth")
    ap.add_argument("--bf16-buckets", action="store_true", help="all-reduce the gradient buckets in bf16 (half the bytes over xGMI; fp32 flat buffer kept)")
    a = ap.parse_args()
    if a.rccl_channels > 0:
        os.environ["NCCL_MAX_NCHANNELS"] = str(a.rccl_channels)
        os.environ["NCCL_MIN_NCHANNELS"] = str(min(a.rccl_channels, int(os.environ.get("NCCL_MIN_NCHANNELS", "1"))))
    if a.bf16_buckets:
        os.environ["LAVT_BF16_BUCKETS"] = "1"

    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(a)                       # never returns

    # stdout carries exactly ONE line, the JSON record.  Libraries write there too (RCCL prints a version banner through C stdio when its first
    # communicator is built, flushed at exit, i.e. after the record): keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and not (a.gpus == 1 and world == 1):
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force = os.environ.get("LAVT_FORCE_COLLECTIVES", "0") == "1"          # dev: run the N>1 code path (SyncBN + bucketed all-reduce) in a 1-rank group
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    import lavt_hip
    from lavt_hip.detweights import det_inputs
    from lavt_hip.engine import TrainStep
    cfg = dict(WORKLOADS[a.workload], name=a.workload)
    if a.dtype is None:
        a.dtype = "fp8" if cfg.get("fp8") else "bf16"
    lavt_hip.set_compute_dtype(a.dtype)
    model = build_model(cfg, device, a.drop_path)
    if world > 1 or force:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)       # train.py:589
    model.train()
    x, l, m, tgt = det_inputs(cfg["batch"], cfg["size"], 20, seed=1234 + rank, frames=cfg.get("frames", 0))
    if cfg.get("one"):                      # token ids + attention mask instead of language features (data/dataset_refer_bert.py:58-81)
        g = torch.Generator("cpu").manual_seed(4321 + rank)
        m = m.squeeze(-1).long()
        l = torch.randint(1000, 30000, m.shape, generator=g) * m
    step = TrainStep(model, x.to(device), l.to(device), m.to(device), tgt.to(device), world=2 if force else world, use_graph=not a.no_graph)
    step.warmup_and_capture()
    if world > 1 and not a.no_graph and not step.captured and os.environ.get("LAVT_DDP_GRAPH", "1") != "0":
        # an eager multi-rank step is bound by ~23 ms of host launch work: reporting it as a scaling number would be misleading.  Fail loudly;
        # LAVT_DDP_GRAPH=0 (or --no-graph) asks for the eager path explicitly.
        print("[bench] hipGraph capture of the multi-rank step failed; refusing to time the eager fallback (set LAVT_DDP_GRAPH=0 to do so on purpose)",
              file=sys.stderr, flush=True)
        os._exit(6)
    if world > 1 and step.captured:
        # first replays of a graph that contains RCCL collectives: bound the damage if a rank never comes back (cannot be tried on the 1-GPU box)
        import threading
        done = threading.Event()

        def _watch():
            if not done.wait(300.0):
                print("[bench] the captured multi-rank step did not complete within 300 s; set LAVT_DDP_GRAPH=0 to launch eagerly", file=sys.stderr, flush=True)
                os._exit(5)
        threading.Thread(target=_watch, daemon=True).start()
        step.step()
        torch.cuda.synchronize()
        done.set()

    for _ in range(a.warmup):
        step.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    def timed_region():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el

    # EXACTLY `steps` steps between the barriers, max over ranks (the contract).  When that region is short (the driver passes --steps 20: 0.2 s) it
    # is repeated and the MEDIAN repetition is reported, so that one scheduling hiccup does not decide the number; `steps` stays what was asked for.
    reps_all = [timed_region()]
    while sum(reps_all) < 2.0 and len(reps_all) < 15:
        reps_all.append(timed_region())
    elapsed = sorted(reps_all)[len(reps_all) // 2]
    loss = float(step.loss)
    opt_ms = None
    if rank == 0 and not a.no_optimizer:
        # the caller's optimizer step (train.py:688-700), reported separately: fused multi-tensor AdamW over the same gradients
        from lavt_hip.optim import FusedAdamW, lavt_param_groups
        opt = FusedAdamW(lavt_param_groups(model), lr=0.0, weight_decay=1e-2, total_steps=1000)      # lr 0: leaves the weights alone
        opt.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            opt.step(check_tables=False)
        e1.record()
        torch.cuda.synchronize()
        opt_ms = e0.elapsed_time(e1) / 10

    if rank == 0:
        images = cfg["batch"] * max(cfg.get("frames", 0), 1) * world * a.steps
        value = images / elapsed
        ms = elapsed / a.steps * 1e3
        train_tflops = 3.0 * FWD_GFLOP_PER_IMAGE[cfg["flops"]] * 1e-3 * value
        out = {
            "metric": ("train frames/sec (384x384 Video-Swin-B LAVT, T=8, fwd+bwd)" if cfg.get("video") else
                       "train images/sec (480x480 Swin-B LAVT, fwd+bwd)" if cfg["variant"] == "base" else "train images/sec (480x480 Swin-T LAVT, fwd+bwd)"),
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": a.workload, "global_batch": cfg["batch"] * world, "image": cfg["size"], "n_l": 20,
                       "parallelism": f"dp{world}", "hip_graph": bool(step.captured), "zero_fill_skipped_values": int(getattr(step, "zero_skip_values", 0)), "drop_path": a.drop_path,
                       "loss": round(loss, 5), "loss_kernel": "fused upsample+CE" if step.fused_loss else "torch CE", "optimizer_ms_separate": None if opt_ms is None else round(opt_ms, 3), "step_tflops_3x_fwd": round(train_tflops, 2),
                       "mfma_frac_of_step": round(train_tflops / world / BF16_DENSE_PEAK_TFLOPS, 4),
                       "timed_repetitions": len(reps_all), "ms_per_step_all_repetitions": [round(e / a.steps * 1e3, 3) for e in reps_all]},
        }
        if world > 1 or force:
            out["config"]["rccl"] = {"max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"), "bf16_buckets": bool(step.buckets.bf16), "ddp_mode": step.buckets.mode,
                                     "gradient_bytes_per_step": step.buckets.bytes_per_step() // (2 if step.buckets.bf16 else 1),
                                     "buckets": len(step.buckets.buckets), "overlappable_bytes": step.buckets.overlappable_bytes() // (2 if step.buckets.bf16 else 1),
                                     "late_bucket_bytes": (step.buckets.bytes_per_step() - step.buckets.overlappable_bytes()) // (2 if step.buckets.bf16 else 1)}
        if world > 1 or force:
            # multi-rank runs: the per-family pass would issue eager collectives on one rank only; the dominant conv kernel is timed alone instead
            # (rank 0, after the timed region, the other ranks idle at the final barrier)
            try:
                out["roofline"] = measure_conv_kernel(device) if a.dtype in ("bf16", "fp8") else {"skipped": "fp32 parity path"}
            except Exception as e:  # noqa: BLE001
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not force:
            try:
                conv = measure_conv_kernel(device) if a.dtype in ("bf16", "fp8") else None
                if a.no_profile:
                    out["roofline"] = conv
                else:
                    roof, prof = profile_step(step, cfg, device)
                    out["roofline"] = roof
                    out["config"]["profile"] = prof
                    if "wmsa_pwam_mfma_frac" in prof:
                        out["config"]["wmsa_pwam_mfma_frac"] = prof["wmsa_pwam_mfma_frac"]
                    out["roofline_conv"] = conv
                if a.dtype in ("bf16", "fp8") and isinstance(out.get("roofline"), dict):
                    mp = measure_attainable_peak(device)
                    out["roofline"]["measured_peak"] = mp
                    best = max([v for v in (mp.get("vendor_hipblaslt"), mp.get("own_gemm_nt")) if isinstance(v, float)], default=None)
                    if best and out["roofline"].get("unit") == "TFLOP/s":
                        out["roofline"]["frac_of_measured_peak"] = round(out["roofline"]["achieved"] / best, 4)
            except Exception as e:  # noqa: BLE001
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                chk = loss_check(model, step, cfg, device)
                if chk is not None:
                    out["config"]["loss_check"] = chk
            except Exception as e:  # noqa: BLE001
                out["config"]["loss_check"] = {"error": f"{type(e).__name__}: {e}"}
            if not a.no_cpu_baseline and not cfg.get("video"):
                out["cpu_baseline"] = cpu_baseline(cfg)
        os.write(record_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
