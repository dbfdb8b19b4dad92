"""Process-wide switches of the HIP path (compute dtype)."""
import contextlib
import os
import warnings

import torch

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
_state = {"dtype": _DTYPES[os.environ.get("LAVT_DTYPE", "fp32").lower()], "fp8": os.environ.get("LAVT_FP8", "0") == "1"}


def set_compute_dtype(dtype) -> None:
    """torch.float32: exact-fp32 MFMA path (parity).  torch.bfloat16: bf16 MFMA, fp32 accumulate (throughput)."""
    if isinstance(dtype, str):
        if dtype.lower() == "fp8":          # bf16 activations / gradients, e4m3 operands for the forward contractions that opt in (set_fp8)
            _state["dtype"], _state["fp8"] = torch.bfloat16, True
            return
        dtype = _DTYPES[dtype.lower()]
    _state["fp8"] = False
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be float32 or bfloat16")
    _state["dtype"] = dtype


_warned_fp16 = [False]


def compute_dtype() -> torch.dtype:
    """bf16 inside torch.autocast('cuda', ...), else the configured dtype.  The reference's own AMP hook is `torch.cuda.amp.autocast()`
    (train.py:452-459), i.e. float16 + GradScaler.  This path has no fp16 kernels -- its low-precision type is bf16 (fp32's exponent range) -- so an
    fp16 autocast region runs the bf16 kernels, with ONE warning per process: `train_one_epoch_ytvos --use_amp` runs unchanged (the GradScaler's
    loss scale passes through harmlessly: bf16 gradients cannot overflow where fp16 ones would, and `scaler.step` unscales the fp32 parameter
    gradients as usual; its scale only ever grows, since no step is skipped).  LAVT_STRICT_FP16_AUTOCAST=1 raises instead;
    LAVT_ALLOW_FP16_AUTOCAST=1 ignores the region and computes in the configured dtype."""
    if torch.is_autocast_enabled():
        adt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
        if adt == torch.bfloat16:
            return torch.bfloat16
        if adt == torch.float16:
            if os.environ.get("LAVT_STRICT_FP16_AUTOCAST", "0") == "1":
                raise RuntimeError("liblavt_hip: called inside torch.autocast(dtype=float16) (the reference's torch.cuda.amp.autocast(), train.py:452) "
                                   "with LAVT_STRICT_FP16_AUTOCAST=1; this path computes in fp32 or bf16 only")
            if os.environ.get("LAVT_ALLOW_FP16_AUTOCAST", "0") == "1":
                # rounds 1-3 meaning of this switch, kept: ignore the fp16 region and compute in the configured dtype (e.g. fp32)
                return _state["dtype"]
            if not _warned_fp16[0]:
                _warned_fp16[0] = True
                warnings.warn("liblavt_hip: torch.autocast(dtype=float16) region (train.py:452) -- this path has no fp16 kernels and computes the region in "
                              "bfloat16 (fp32 accumulation; no loss scaling needed, a GradScaler is harmless)", RuntimeWarning, stacklevel=2)
            return torch.bfloat16
    return _state["dtype"]


def fp8_enabled() -> bool:
    """BASELINE.json configs[4]: e4m3 weights / activations on the CDNA4 fp8 MFMA for the forward contractions (bf16 everywhere else)."""
    return _state["fp8"] and _state["dtype"] == torch.bfloat16


@contextlib.contextmanager
def use_dtype(dtype):
    prev = dict(_state)
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _state.update(prev)
