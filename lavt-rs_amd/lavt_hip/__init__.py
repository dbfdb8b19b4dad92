"""lavt_hip: host side of the MI355X-native LAVT hot path (ctypes over liblavt_hip.so).

Importing `lavt_hip.ops` (or anything under `lib/`) loads the shared library and fails loudly when
it has not been built; `lavt_hip.rowmaps`, `lavt_hip.detweights` and `lavt_hip.runtime` are plain
host logic and import without it.
"""
from .runtime import compute_dtype, fp8_enabled, set_compute_dtype, use_dtype  # noqa: F401
