"""world_size-2 `gloo` tests (CPU) of the data-parallel pieces: bucketed, backward-overlapped gradient all-reduce
(lavt_hip.ddp.GradBuckets) incl. parameters that never receive a gradient (the reference needs
find_unused_parameters=True for layers.3.res_gate.*, train.py:592)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.unused = nn.Linear(16, 16, bias=False)       # never reaches the loss
        self.c = nn.Linear(16, 4)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, bucket_mib, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavt_hip.ddp import GradBuckets
    torch.manual_seed(0)
    model = Toy()
    if rank == 1:                                  # ranks start different: the constructor must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    gb = GradBuckets(model, bucket_mib=bucket_mib)
    ref = Toy()
    torch.manual_seed(0)
    ref = Toy()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q), f"{n} not broadcast"
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    for step in range(2):                          # two steps: zero() must reset everything
        gb.zero()
        model(xs[rank]).pow(2).mean().backward()
        gb.finish()
    # reference: average of the per-rank gradients, computed locally
    grads = []
    for r in range(world):
        ref.zero_grad()
        ref(xs[r]).pow(2).mean().backward()
        grads.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()})
    for n, p in model.named_parameters():
        want = sum(g[n] for g in grads) / world
        assert p.grad is not None and torch.allclose(p.grad, want, atol=1e-6), n
    assert float(model.unused.weight.grad.abs().max()) == 0.0
    assert len(gb.buckets) >= (2 if bucket_mib < 1e-3 else 1)
    out[rank] = True
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mib", [64.0, 0.0005])
def test_grad_buckets_world2(bucket_mib):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, bucket_mib, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}
