import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lavt-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """LAVT_TEST_POISON=1 (GPU box): before every GPU test the caching allocator's free blocks are filled with NaN bit patterns, so a kernel that reads a
    `torch.empty` buffer it was supposed to write first (scratch, partial records, padded rows) shows up as NaN in the comparison instead of passing on
    whatever the previous test left there.  Off by default (it adds ~1 GB of fills per test)."""
    if os.environ.get("LAVT_TEST_POISON") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            blocks = []
            try:
                for _ in range(4):
                    blocks.append(torch.full((64 << 20,), float("nan"), dtype=torch.float32, device="cuda:0"))          # 4 x 256 MB
            except RuntimeError:
                pass
            torch.cuda.synchronize()
            del blocks
    yield
