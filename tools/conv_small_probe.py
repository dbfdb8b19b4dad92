#!/usr/bin/env python3
"""Where does the time of the decoder's small-pixel-count convolutions go?  Forward 3x3 convolution at M = 7200 / 1800 pixels against a plain GEMM of
the same M x N x K, hot (same operands every launch) and cold (600 MB fill between launches), under the tile / ring-depth switches.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops, _capi as K
dev, bf = "cuda:0", torch.bfloat16
lavt_hip.set_compute_dtype(bf)

def graph_time(fns, reps=5):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

flush = torch.empty(600 * 1024 * 1024 // 4, device=dev)
fl = lambda: flush.fill_(1.0)
tf = graph_time([fl] * 8) / 8

def hot_cold(fn, n=16):
    hot = graph_time([fn] * n) / n
    cold = graph_time([f for _ in range(8) for f in (fl, fn)]) / 8 - tf
    return hot, cold

def setenv(**kw):
    for k, v in kw.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    K.lib.lavt_tuning_reload()

shapes = ((2, 120, 512, 0, 512), (2, 120, 512, 128, 512), (2, 60, 512, 0, 512), (2, 60, 512, 256, 512), (2, 30, 512, 0, 512), (2, 30, 1024, 512, 512))
only = os.environ.get("PROBE_SHAPES")
if only: shapes = tuple(shapes[int(i)] for i in only.split(","))
for (B, H, C1, C2, Cout) in shapes:
    M, Cin = B * H * H, C1 + C2
    x1 = torch.randn(M, C1, device=dev).to(bf)
    x2 = torch.randn(M, C2, device=dev).to(bf) if C2 else None
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
    PAD = int(os.environ.get("PROBE_PAD", "0"))        # row padding (elements) of the plain GEMM's operands: L2 channel spread
    LDP = 9 * Cin + PAD
    Ap = torch.randn(M, LDP, device=dev).to(bf)
    Wp = torch.randn(Cout, LDP, device=dev).to(bf)
    Cc = torch.empty(M, Cout, device=dev, dtype=bf)
    gf = 2.0 * M * Cout * 9 * Cin * 1e-9
    def conv():
        with torch.no_grad():
            return ops.conv3x3(x1, x2, w, B, H, H)
    setenv(LAVT_GEMM_PIPE=0)
    y_ref = conv().float()
    Wc = w.detach().to(bf).permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()   # k-contiguous [Cout][tap][Cin]: plain GEMM against a k-major read of it = data-gradient shape
    def plain():
        ops.gemm_nt(bf, M, Cout, 9 * Cin, Ap, LDP, Wp, LDP, Cc, Cout)
    Wpk = ops.weights.get(w, bf, "conv3")
    dy = torch.randn(M, Cout, device=dev).to(bf)
    dx = torch.empty(M, C1, device=dev, dtype=bf)
    def dgrad():          # data gradient of the first source (k-major read of the packed weight), as ops._ConvTaps.backward issues it
        sp, over_ch = ops._conv_split(bf, M, C1, Cout, Cout, 0, 9, None, K.ACT_NONE)
        if sp and over_ch:
            parts = torch.empty(sp, M, C1, dtype=torch.float32, device=dev)
            ops.gemm_nt(bf, M, C1, 9 * (Cout // sp), dy, Cout, Wpk, 9 * Cin, parts, C1, conv=(H, H, Cout, 1, 1, 1, 3, 3), b_kmajor=True, b_tap_stride=Cin, b_off=0,
                        batch=sp, strideC=M * C1, c_f32=True, conv_kc_split=Cout // sp)
            K.check(K.lib.lavt_splitk_reduce(K.dt(bf), K.ptr(parts), sp, M, C1, K.ptr(dx), C1, K.stream()))
        elif sp:
            parts = torch.empty(sp, M, C1, dtype=torch.float32, device=dev)
            ops.gemm_nt(bf, M, C1, (9 // sp) * Cout, dy, Cout, Wpk, 9 * Cin, parts, C1, conv=(H, H, Cout, 1, 1, 1, 3, 3), b_kmajor=True, b_tap_stride=Cin, b_off=0,
                        batch=sp, strideC=M * C1, c_f32=True, conv_tap_split=9 // sp)
            K.check(K.lib.lavt_splitk_reduce(K.dt(bf), K.ptr(parts), sp, M, C1, K.ptr(dx), C1, K.stream()))
        else:
            ops.gemm_nt(bf, M, C1, 9 * Cout, dy, Cout, Wpk, 9 * Cin, dx, C1, conv=(H, H, Cout, 1, 1, 1, 3, 3), b_kmajor=True, b_tap_stride=Cin, b_off=0)
        return dx
    d_ref = dgrad().float().clone()
    gfd = 2.0 * M * C1 * 9 * Cout * 1e-9
    print(f"--- M={M} Cin={C1}+{C2} Cout={Cout}  ({gf:.1f} GFLOP; 0.40 of peak = {gf / 1.0:.1f} us)", flush=True)
    variants = (("v2 default", {"LAVT_GEMM_PIPE": 0}, None), ("pipe default", {"LAVT_GEMM_PIPE": 2}, None),
                ("v2 nosplit", {"LAVT_GEMM_PIPE": 0}, 0), ("pipe nosplit", {"LAVT_GEMM_PIPE": 2}, 0),
                ("v2 128 st4", {"LAVT_GEMM_PIPE": 0, "LAVT_GEMM_TILE": 128, "LAVT_GEMM_STAGES": 4}, 0),
                ("pipe 128 st4", {"LAVT_GEMM_PIPE": 3, "LAVT_GEMM_TILE": 128, "LAVT_GEMM_STAGES": 4}, 0),
                ("pipe 128 st2", {"LAVT_GEMM_PIPE": 3, "LAVT_GEMM_TILE": 128, "LAVT_GEMM_STAGES": 2}, 0),
                ("v2 256", {"LAVT_GEMM_PIPE": 0, "LAVT_GEMM_TILE": 512}, 0), ("pipe 256", {"LAVT_GEMM_PIPE": 2, "LAVT_GEMM_TILE": 512}, 0))
    variants = variants + tuple((f"kc-split {k}", {"LAVT_GEMM_PIPE": 2}, None, k) for k in ("0", "2", "4", "8", "auto")) if M <= 2048 else variants
    if os.environ.get("PROBE_FULL"):
        variants = variants + (("tile64", {"LAVT_GEMM_PIPE": 0, "LAVT_GEMM_TILE": 64}, 0), ("split-all", {"LAVT_GEMM_PIPE": 2}, 1 << 20))
    if os.environ.get("PROBE_VARIANTS"):
        variants = tuple(v for v in variants if v[0] in os.environ["PROBE_VARIANTS"].split(",") or v[0] == "v2 default")
    for tag, env, rows, *kc in variants:
        setenv(LAVT_GEMM_TILE=None, LAVT_GEMM_STAGES=None, LAVT_GEMM_PIPE=None)
        setenv(**env)
        old = ops._CONV_SPLIT_MAX_ROWS
        if rows is not None: ops._CONV_SPLIT_MAX_ROWS = rows
        old_kc = ops._CONV_KC_SPLITS
        if kc: ops._CONV_KC_SPLITS = kc[0]
        try:
            err = float((conv().float() - y_ref).abs().max() / y_ref.abs().max())
            plain(); c_chk = Cc.float().clone()
            ch, cc = hot_cold(conv)
            ph, pc = hot_cold(plain)
            derr = float((dgrad().float() - d_ref).abs().max() / d_ref.abs().max())
            dh, dc = hot_cold(dgrad)
            if tag == "v2 default": c_ref = c_chk
            perr = float((c_chk - c_ref).abs().max() / c_ref.abs().max())
            print(f"  {tag:14s} conv hot {ch:6.1f} cold {cc:6.1f} us ({gf / ch * 1e3 / 2500:.2f} of peak, diff {err:.1e}) | plain GEMM hot {ph:6.1f} cold {pc:6.1f} us ({gf / ph * 1e3 / 2500:.2f}, diff {perr:.1e}) | dgrad hot {dh:6.1f} cold {dc:6.1f} ({gfd / dh * 1e3 / 2500:.2f}, diff {derr:.1e})", flush=True)
        except Exception as e:
            print(f"  {tag:14s} failed: {e}", flush=True)
        ops._CONV_SPLIT_MAX_ROWS = old
        ops._CONV_KC_SPLITS = old_kc
    setenv(LAVT_GEMM_TILE=None, LAVT_GEMM_STAGES=None, LAVT_GEMM_PIPE=None)
