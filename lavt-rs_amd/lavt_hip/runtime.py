"""Process-wide switches of the HIP path (compute dtype)."""
import contextlib
import os

import torch

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
_state = {"dtype": _DTYPES[os.environ.get("LAVT_DTYPE", "fp32").lower()]}


def set_compute_dtype(dtype) -> None:
    """torch.float32: exact-fp32 MFMA path (parity).  torch.bfloat16: bf16 MFMA, fp32 accumulate (throughput)."""
    if isinstance(dtype, str):
        dtype = _DTYPES[dtype.lower()]
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be float32 or bfloat16")
    _state["dtype"] = dtype


def compute_dtype() -> torch.dtype:
    """bf16 inside torch.autocast('cuda', dtype=bfloat16) (the caller's AMP hook, train.py:452), else the configured dtype."""
    if torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16:
        return torch.bfloat16
    return _state["dtype"]


@contextlib.contextmanager
def use_dtype(dtype):
    prev = _state["dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _state["dtype"] = prev
