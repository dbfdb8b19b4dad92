"""ctypes binding of liblavt_hip.so (C ABI declared in include/lavt_hip.h).

There is NO fallback: if the shared library is missing the import raises, and every call that
returns a non-zero status raises RuntimeError with the library's error string.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAVT_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "liblavt_hip.so")          # (override: A/B of two builds on one box)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C lavt-rs_amd/csrc` (or __graft_entry__.build()). "
        "The LAVT HIP path has no CPU/eager fallback.")
_cdll = C.CDLL(LIB_PATH)


class _Profiler:
    """In-process kernel timing for bench.py: when enabled, every C-ABI launch is bracketed by two HIP events recorded on the
    stream the kernel is launched on, and tagged with the entry point, the caller's scope label ("wmsa", "pwam", ...: lavt_hip._capi.scope) and a
    shape / algorithmic-work note left by the host wrapper.  Off by default (one attribute test per launch); never on inside a hipGraph capture."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (entry point, scope, note, start event, end event)
        self.label = "other"
        self.note = None

    def start(self):
        self.records, self.enabled, self.note = [], True, None

    def stop(self):
        """-> list of (entry point, scope, note, microseconds)"""
        self.enabled = False
        torch.cuda.synchronize()
        out = [(n, s, d, e0.elapsed_time(e1) * 1e3) for n, s, d, e0, e1 in self.records]
        self.records = []
        return out


prof = _Profiler()


class scope:
    """`with scope("wmsa"):` labels the launches issued inside (forward); autograd Functions decorated with @scoped carry the label of their
    forward into their backward."""

    def __init__(self, label):
        self.label = label

    def __enter__(self):
        self.prev, prof.label = prof.label, self.label

    def __exit__(self, *exc):
        prof.label = self.prev
        return False


def scoped(fn_cls):
    """class decorator for torch.autograd.Function: backward launches inherit the scope label that was current in forward"""
    fwd, bwd = fn_cls.forward, fn_cls.backward

    def forward(ctx, *a):
        ctx._lavt_scope = prof.label
        return fwd(ctx, *a)

    def backward(ctx, *g):
        with scope(getattr(ctx, "_lavt_scope", "other")):
            return bwd(ctx, *g)

    fn_cls.forward, fn_cls.backward = staticmethod(forward), staticmethod(backward)
    return fn_cls


class _Lib:
    """The CDLL behind a thin proxy: `lib.lavt_x(...)` is the ctypes function itself unless the profiler is on."""

    def __init__(self, cdll):
        self._c = cdll

    def __getattr__(self, name):
        fn = getattr(self._c, name)

        def call(*args):
            if not prof.enabled:
                return fn(*args)
            note, prof.note = prof.note, None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            prof.records.append((name, prof.label, note, e0, e1))
            return rc
        setattr(self, name, call)
        return call


lib = _Lib(_cdll)

F32, BF16, FP8 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_GELU_D, ACT_STORED = 0, 1, 2, 3, 4, 5
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmNT(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
        ("A", vp), ("lda", i64), ("strideA", i64), ("A2", vp), ("lda2", i64), ("a_split", i32), ("a_rowmap", vp),
        ("conv_h", i32), ("conv_w", i32), ("conv_kc", i32), ("conv_flip", i32),
        ("B", vp), ("ldb", i64), ("strideB", i64), ("b_kmajor", i32), ("b_tap_stride", i64),
        ("alpha", f32), ("bias", vp), ("strideBias", i64), ("row_scale", vp), ("strideRowScale", i64), ("row_scale_div", i32), ("act", i32),
        ("Cpre", vp), ("ldcpre", i64), ("R", vp), ("ldr", i64),
        ("C", vp), ("ldc", i64), ("strideC", i64), ("C2", vp), ("ldc2", i64), ("c_split", i32), ("c_rowmap", vp),
        ("c_f32", i32), ("zeros", vp), ("epi_lds", i32), ("conv_d", i32), ("conv_kd", i32), ("conv_kh", i32), ("conv_kw", i32),
        ("dact_pre", vp), ("lddact", i64), ("dact", i32), ("deq_a", vp), ("deq_b", vp), ("epi_wide", i32),
        ("mul", vp), ("ldmul", i64), ("res_first", i32), ("conv_tap_split", i32),
        ("ln_wsum", vp), ("ln_mean", vp), ("ln_rstd", vp), ("ln_eps", f32), ("colstats", vp), ("conv_kc_split", i32),
    ]


class DtableJob(C.Structure):
    """lavt_dtable_job_t: the table-gradient binning of one attention-backward launch (chained form)"""
    _fields_ = [("slab", vp), ("part", vp), ("slab_ld", i32), ("wd", i32), ("wh", i32), ("ww", i32), ("nwin", i32), ("N", i32), ("heads", i32),
                ("rows_per_block", i32), ("win_per_group", i32), ("gx", i32), ("gz", i32)]


class GemmTN(C.Structure):
    _fields_ = [
        ("dtype", i32), ("I", i32), ("J", i32), ("K", i32), ("batch", i32),
        ("A", vp), ("lda", i64), ("strideA", i64), ("a_rowmap", vp), ("a_rowscale", vp), ("a_rowscale_div", i32),
        ("B", vp), ("ldb", i64), ("strideB", i64), ("B2", vp), ("ldb2", i64), ("b_split", i32), ("b_rowmap", vp),
        ("conv_h", i32), ("conv_w", i32), ("conv_kc", i32),
        ("alpha", f32), ("C", vp), ("ldc", i64), ("strideC", i64), ("c_conv_permute", i32), ("split_k", i32),
        ("colsum", vp), ("strideColsum", i64), ("zeros", vp), ("a_rowscale_binary", i32), ("accumulate", i32),
        ("conv_d", i32), ("conv_kd", i32), ("conv_kh", i32), ("conv_kw", i32),
        ("partials", vp), ("partials_floats", i64), ("colsum_atomic", i32), ("a_src_rows", i64), ("b_src_rows", i64),
    ]


# every symbol include/lavt_hip.h declares, with its argument types (tests check this list against the header)
_PROTOTYPES = {
    "lavt_abi_version": [],
    "lavt_tuning_reload": [],
    "lavt_gemm_nt_colstats_plan": [C.POINTER(GemmNT), C.POINTER(i32)],
    "lavt_colstats_finish_blocks": [vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, f32, vp],
    "lavt_window_attn_bwd_chained": [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, f32, C.POINTER(DtableJob), C.POINTER(DtableJob), vp],
    "lavt_attn_dtable_run": [C.POINTER(DtableJob), vp],
    "lavt_lang_mask": [vp, i32, vp, vp, i32, i32, i32, vp],
    "lavt_droppath_factors": [vp, vp, vp, i32, i32, vp],
    "lavt_droppath_draw": [vp, vp, vp, i32, i32, vp],
    "lavt_conv3x3_wgrad_ws": [i32, i32, i32, i32, i32, i32],
    "lavt_gemm_tn_grouped_ln": [C.POINTER(GemmTN), i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp],
    "lavt_gemm_tn_grouped_sk_ws": [C.POINTER(GemmTN), i32],
    "lavt_gemm_tn_grouped_sk": [C.POINTER(GemmTN), i32, vp, i64, vp],
    "lavt_conv3x3_wgrad": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp, i32, vp, vp],
    "lavt_conv3x3_wgrad_f8_ok": [i32, i32, i32, i32, i32, i32],
    "lavt_conv3x3_wgrad_f8": [vp, i64, vp, vp, i64, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, i32, vp, vp],
    "lavt_gemm_nt": [C.POINTER(GemmNT), vp],
    "lavt_gemm_tn": [C.POINTER(GemmTN), vp],
    "lavt_splitk_reduce": [i32, vp, i32, i64, i32, vp, i64, vp],
    "lavt_gemm_tn_pieces": [C.POINTER(GemmTN)],
    "lavt_gemm_tn_grouped": [C.POINTER(GemmTN), i32, vp],
    "lavt_window_attn_fwd": [i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "lavt_window_attn_bwd": [i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "lavt_window_attn_bwd_pieces": [i32, i32, i32, i32, i32],
    "lavt_attn_dtable_finish_multi": [vp, i32, i32, i32, vp],
    "lavt_attn_dtable_finish_multi_compact": [vp, i32, i32, i32, vp],
    "lavt_attn_uses_table": [i32, i32],
    "lavt_window_attn_bwd_ws": [i32, i32, i32, i32, i32, i32, i32, i32],
    "lavt_relpos_expand": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lavt_relpos_reduce": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lavt_attn_softmax_fwd": [i32, vp, vp, i32, vp, i32, vp, i64, i32, i32, i32, i32, vp],
    "lavt_attn_softmax_bwd": [i32, vp, vp, i64, i32, i32, vp],
    "lavt_attn_dbias_sum": [i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "lavt_layernorm_fwd": [i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "lavt_layernorm_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp],
    "lavt_layernorm_bwd_blocks": [i32, i32, i32],
    "lavt_layernorm_bwd_partial": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp],
    "lavt_layernorm_bwd_partial_xn": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp],
    "lavt_layernorm_bwd_partial_xn_dtable": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, C.POINTER(DtableJob), vp],
    "lavt_layernorm_bwd_xn": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, vp],
    "lavt_reduce_partials_multi": [vp, i32, i32, vp],
    "lavt_reduce_partials_column_blocks": [i32],
    "lavt_colstats": [i32, vp, vp, vp, vp, i64, i32, i32, i32, vp],
    "lavt_syncbn_combine": [vp, i32, f32, f32, vp, vp, vp, vp, f32, i32, vp],
    "lavt_colstats_meanrstd": [i32, vp, vp, vp, vp, i64, i32, i32, i32, f32, vp, vp, f32, vp],
    "lavt_stats_finalize": [vp, vp, f32, f32, vp, vp, vp, vp, f32, i32, vp],
    "lavt_norm_apply": [i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp],
    "lavt_norm_bwd_stats": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, i32, i32, i32, vp],
    "lavt_norm_bwd_apply": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, f32, vp, vp, i32, i32, i32, vp],
    "lavt_act_bwd": [i32, i32, vp, vp, vp, i64, vp],
    "lavt_bert_embed_fwd": [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_bert_embed_bwd": [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_dropout": [i32, vp, vp, f32, vp, vp, i64, vp],
    "lavt_gate_fwd": [i32, vp, vp, vp, vp, i64, vp],
    "lavt_gate_bwd": [i32, vp, vp, vp, vp, vp, vp, i64, vp],
    "lavt_ln_fold": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "lavt_wmsa_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp],
    "lavt_wmsa_fwd_rider": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp, i64, vp],
    "lavt_pwam_words_fwd": [vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "lavt_pwam_words_fwd_moments": [vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "lavt_pwam_lang_fwd_records": [vp, i64, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "lavt_pwam_words_records": [i32, i32, i32],
    "lavt_pwam_words_bwd": [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_pwam_q_parts": [i32],
    "lavt_pwam_mix": [i32, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "lavt_pwam_lang_fwd": [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "lavt_pwam_lang_bwd1": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_pwam_mix1": [vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, vp],
    "lavt_pwam_mix1_records": [i32, i32, i32],
    "lavt_pwam_lang_bwd1_records": [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_pwam_lang_bwd2": [vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, vp],
    "lavt_rowsoftmax_fwd": [i32, vp, vp, i64, i32, i32, vp],
    "lavt_rowsoftmax_bwd": [i32, vp, vp, vp, i64, i32, i32, vp],
    "lavt_bilinear_fwd": [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lavt_bilinear_fwd_q8": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lavt_norm_apply_q8": [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp],
    "lavt_norm_bwd_apply_amax": [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, f32, vp, vp, vp, i32, i32, i32, vp],
    "lavt_bilinear_bwd": [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lavt_logits_up_fwd": [i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "lavt_logits_up_bwd": [i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "lavt_unpack_conv_grad": [vp, vp, i32, i32, i32, vp],
    "lavt_adamw_step": [vp, vp, i32, vp, f32, f32, vp],
    "lavt_adamw_step_chunks": [vp, vp, vp, i32, vp, f32, f32, vp],
    "lavt_adamw_chunk_elems": [],
    "lavt_ln_fold_multi": [vp, i32, vp],
    "lavt_upsample_ce_fwd": [i32, vp, vp, f32, f32, vp, i64, vp, i32, i32, i32, i32, i32, vp],
    "lavt_upsample_ce_bwd": [i32, vp, vp, f32, f32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "lavt_upsample_dice_fwd": [i32, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, vp],
    "lavt_upsample_dice_bwd": [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "lavt_fp8_quantize": [i32, vp, vp, i64, vp, vp, vp],
    "lavt_fp8_advance": [vp, vp, i32, vp],
    "lavt_fp8_quantize_weight": [vp, vp, vp, i32, i32, i32, vp],
    "lavt_fp8_quantize_weight_t": [vp, vp, vp, i32, i32, i32, vp],
    "lavt_fp8_quantize_current": [i32, vp, vp, i64, vp, vp],
    "lavt_cls_head_fwd": [i32, vp, vp, vp, vp, i64, i32, vp],
    "lavt_cls_head_bwd": [i32, vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "lavt_cls_head_bwd_partial": [i32, vp, vp, vp, vp, vp, vp, i64, i32, vp],
    "lavt_cls_head_bwd_blocks": [i32, i64, i32],
    "lavt_im2col4": [i32, vp, vp, i32, i32, i32, vp],
    "lavt_col2im4": [i32, vp, vp, i32, i32, i32, vp],
    "lavt_cast": [i32, vp, i32, vp, i64, vp],
    "lavt_nchw_to_nhwc": [i32, vp, i32, vp, i32, i32, i32, vp],
    "lavt_nhwc_to_nchw": [i32, vp, i32, vp, i32, i32, i32, vp],
    "lavt_pack_conv3x3": [vp, i32, vp, i32, i32, i32, vp],
    "lavt_cast_multi": [vp, i32, i32, vp],
}
for _name, _args in _PROTOTYPES.items():
    _fn = getattr(_cdll, _name)          # AttributeError here = header/library mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = C.c_int
EXPECTED_ABI = 7          # the ctypes struct layouts and prototypes in this file were written for this lavt_abi_version()
if _cdll.lavt_abi_version() != EXPECTED_ABI:
    raise ImportError(f"{LIB_PATH} reports ABI v{_cdll.lavt_abi_version()} but lavt_hip/_capi.py binds ABI v{EXPECTED_ABI}: rebuild the library "
                      "(`make -C lavt-rs_amd/csrc`) -- a mismatch would make the kernels read past the caller's parameter structs")
_cdll.lavt_last_error.restype = C.c_char_p
_cdll.lavt_window_attn_bwd_ws.restype = C.c_int64
_cdll.lavt_conv3x3_wgrad_ws.restype = C.c_int64
_cdll.lavt_gemm_tn_grouped_sk_ws.restype = C.c_int64
_cdll.lavt_last_error.argtypes = []
for _name in ("lavt_last_error", "lavt_window_attn_bwd_ws", "lavt_attn_uses_table", "lavt_abi_version", "lavt_layernorm_bwd_blocks", "lavt_window_attn_bwd_pieces", "lavt_gemm_tn_pieces", "lavt_pwam_q_parts", "lavt_pwam_words_records", "lavt_pwam_mix1_records", "lavt_adamw_chunk_elems", "lavt_tuning_reload", "lavt_conv3x3_wgrad_ws", "lavt_conv3x3_wgrad_f8_ok", "lavt_gemm_tn_grouped_sk_ws", "lavt_gemm_nt_colstats_plan", "lavt_cls_head_bwd_blocks", "lavt_reduce_partials_column_blocks"):      # queries, not launches: never timed
    setattr(lib, _name, getattr(_cdll, _name))

EXPORTED = tuple(_PROTOTYPES) + ("lavt_last_error",)


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"liblavt_hip: rc={rc}: {lib.lavt_last_error().decode()}")


def dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"liblavt_hip computes in float32 or bfloat16, not {t}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("liblavt_hip operates on GPU memory only (got a CPU tensor); there is no CPU fallback")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
