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


def _bf16_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LAVT_BF16_BUCKETS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavt_hip.ddp import GradBuckets
    torch.manual_seed(0)
    model = Toy()
    gb = GradBuckets(model, bucket_mib=0.0005)
    assert gb.bf16
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    gb.zero()
    model(xs[rank]).pow(2).mean().backward()
    gb.finish()
    ref = Toy()
    torch.manual_seed(0)
    ref = Toy()
    grads = []
    for r in range(world):
        ref.zero_grad()
        ref(xs[r]).pow(2).mean().backward()
        grads.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()})
    for n, p in model.named_parameters():
        want = sum(g[n].to(torch.bfloat16).float() for g in grads) / world          # each rank's contribution rounded to bf16 before the sum
        assert p.grad.dtype == torch.float32 and torch.allclose(p.grad, want, rtol=2e-2, atol=1e-6), n
        exact = sum(g[n] for g in grads) / world
        assert float((p.grad - exact).abs().max()) <= 1e-2 * float(exact.abs().max()) + 1e-7, n
    out[rank] = True
    dist.destroy_process_group()


def test_bf16_grad_buckets_world2():
    """LAVT_BF16_BUCKETS=1: buckets are reduced as bf16 copies and written back into the fp32 flat buffer: gradients = the mean of the ranks' bf16-rounded
    gradients (within 1 % of the exact mean), .grad stays fp32"""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bf16_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


@pytest.mark.parametrize("bucket_mib", [64.0, 0.0005])
def test_grad_buckets_world2(bucket_mib):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, bucket_mib, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def _disagree_worker(rank, world, port, out):
    """step 1 is data-dependent: rank 0 never reports weight 1, rank 1 never reports weight 2.  Decided per rank the relayout would give the ranks
    different bucket boundaries (all-reduce chunks of different content); the union decides, so both end with the same layout and the exchange of
    the following steps is the plain average."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavt_hip.ddp import GradBuckets
    torch.manual_seed(0)
    net = nn.Sequential(*[nn.Linear(8, 8, bias=False) for _ in range(4)])
    ps = list(net.parameters())
    gb = GradBuckets(net, bucket_mib=2 * 64 * 4 / (1 << 20))            # two weights per bucket
    g = [torch.full((8, 8), float(10 * rank + i)) for i in range(4)]
    for step in range(3):
        gb.zero()
        skip = {0: 1, 1: 2}[rank] if step == 0 else None
        for i in reversed(range(4)):
            if i != skip:
                ps[i].grad.copy_(g[i])
                gb._on_grad(ps[i])
        gb.finish()
        if step > 0:
            assert gb.late_bucket is not None and {i for i in range(4) if gb.bucket_of[ps[i]] == gb.late_bucket} == {1, 2}
            for i in range(4):
                assert torch.allclose(ps[i].grad, torch.full((8, 8), float(i + 5.0))), (step, i)          # mean of (i, 10 + i)
    out[rank] = [list(b) for b in gb.buckets]
    dist.destroy_process_group()


def test_grad_buckets_relayout_agrees_across_ranks_world2():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_disagree_worker, args=(2, port, out), nprocs=2, join=True)
        res = dict(out)
        assert res[0] == res[1] and len(res[0]) == 2


# ------------------------------------------------------------------------------------------------ SyncBN protocol, two real processes
class _CpuBnKernels:
    """CPU stand-in for the local BatchNorm + ReLU passes of csrc/norm.hip (same interface as lavt_hip.ops._HipBnKernels, same definitions:
    sum / centred second moment, biased variance for normalisation and unbiased for the running estimate, ReLU mask from the saved output)."""

    @staticmethod
    def stats(x):
        return torch.stack([x.sum(0), ((x - x.mean(0)) ** 2).sum(0)]).view(2, 1, -1)

    @staticmethod
    def finalize(s, count, eps, running_mean, running_var, momentum):
        mean, var = s[0] / count, s[1] / count
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean)
            running_var.mul_(1 - momentum).add_(momentum * var * (count / max(count - 1.0, 1.0)))
        return mean, torch.rsqrt(var + eps)

    @staticmethod
    def apply(x, mean, rstd, gamma, beta):
        return torch.relu((x - mean) * rstd * gamma + beta)

    @staticmethod
    def bwd_stats(dy, x, y, mean, rstd, gamma, beta, out=None):
        g = dy * (y > 0)
        s0, s1 = g.sum(0), (g * (x - mean) * rstd).sum(0)
        if out is not None:                     # the product's form: accumulate into the (zeroed) gradient sinks of beta / gamma
            out[0].add_(s0)
            out[1].add_(s1)
            return out
        return s0, s1

    @staticmethod
    def bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count):
        g = dy * (y > 0)
        xh = (x - mean) * rstd
        return gamma * rstd * (g - s[0] / count - xh * s[1] / count)


class _CpuBnKernelsShipping(_CpuBnKernels):
    """the same local passes + the two entry points that select the SHIPPING multi-rank branches of _BatchNormRelu (ops.py: syncbn_gather +
    combine_finalize forward; local sums into the gradient sinks + one stacked all-reduce backward).  combine_finalize restates
    csrc/norm.hip:syncbn_combine_kernel: allst [world, 2, 1, C] = every rank's (sum, centred M2) over `rows` rows."""
    fused_sinks = True

    @staticmethod
    def combine_finalize(allst, rows, eps, running_mean, running_var, momentum):
        world = allst.shape[0]
        a = allst.reshape(world, 2, -1)
        n = float(rows * world)
        gmean = a[:, 0].sum(0) / n
        m2 = (a[:, 1] + rows * (a[:, 0] / rows - gmean) ** 2).sum(0)
        var = m2 / n
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * gmean)
            running_var.mul_(1 - momentum).add_(momentum * var * (n / max(n - 1.0, 1.0)))
        return gmean, torch.rsqrt(var + eps)



def _syncbn_worker(rank, world, port, out, shipping=False, use_sinks=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavt_hip import ops
    kern = _CpuBnKernelsShipping if shipping else _CpuBnKernels
    R, C = 37, 6
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(world * R, C, generator=g, dtype=torch.float64) * 0.5 + 3.0 * torch.arange(1, C + 1)        # large means: no E[x^2]-E[x]^2
    dy_all = torch.randn(world * R, C, generator=g, dtype=torch.float64)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64))
    beta = 0.1 * torch.randn(C, generator=g, dtype=torch.float64)
    # this rank's shard through the product's autograd Function: real all_gather_into_tensor / all_reduce between the two processes
    x = x_all[rank * R:(rank + 1) * R].clone().float().requires_grad_(True)
    gm, bt = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    if use_sinks:                               # the step harness' fused accumulation: .grad = zeroed views the op accumulates into, no gradient tensors returned
        gm.grad, bt.grad = torch.zeros(C), torch.zeros(C)
        ops.sinks.set([gm, bt])
    try:
        y = ops._BatchNormRelu.apply(x, gm, bt, rm, rv, True, 0.1, 1e-5, dist.group.WORLD, kern)
        y.backward(dy_all[rank * R:(rank + 1) * R].float())
    finally:
        ops.sinks.clear()
    # single-process BatchNorm over the concatenated batch (float64)
    bn = nn.BatchNorm1d(C, eps=1e-5, momentum=0.1).double().train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xr = x_all.clone().requires_grad_(True)
    yr = torch.relu(bn(xr))
    yr.backward(dy_all)
    sl = slice(rank * R, (rank + 1) * R)
    assert torch.allclose(y.detach().double(), yr[sl].detach(), atol=2e-5), "forward"
    assert torch.allclose(rm.double(), bn.running_mean, atol=1e-5) and torch.allclose(rv.double(), bn.running_var, atol=1e-5), "running statistics"
    assert torch.allclose(x.grad.double(), xr.grad[sl], atol=2e-5), "dx"
    # parameter gradients stay local sums (DDP averages them afterwards): their sum over the ranks is the single-process gradient
    tot = torch.stack([gm.grad, bt.grad])
    dist.all_reduce(tot)
    assert torch.allclose(tot[0].double(), bn.weight.grad, atol=1e-4) and torch.allclose(tot[1].double(), bn.bias.grad, atol=1e-4), "dgamma / dbeta"
    # eval mode never communicates and uses the running estimates
    ye = ops._BatchNormRelu.apply(x.detach(), gm.detach(), bt.detach(), rm, rv, False, 0.1, 1e-5, None, kern)
    assert torch.allclose(ye.double(), torch.relu(bn.eval()(x_all[sl])).detach(), atol=2e-5)
    out[rank] = True
    dist.destroy_process_group()


def test_syncbn_exchange_world2():
    """lavt_hip.ops._BatchNormRelu with a process group: statistics exchanged with ONE all_gather_into_tensor forward and ONE all_reduce backward
    between two processes == BatchNorm over the concatenated batch (output, running statistics, dx, dgamma / dbeta); local passes on CPU."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_syncbn_worker, args=(2, port, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


@pytest.mark.parametrize("use_sinks", [False, True])
def test_syncbn_shipping_branch_world2(use_sinks):
    """The branch the GPU product takes when world > 1 (ops._BatchNormRelu with a kernel set that has combine_finalize + fused_sinks): forward =
    syncbn_gather of every rank's (sum, centred M2) pair + ONE combination of the pairs (the stand-in restates lavt_syncbn_combine: counts
    rows * world, layout [world, 2, 1, C]); backward = local sums into the parameters' gradient sinks (or fresh buffers without sinks) + one
    all-reduce of a stacked copy.  Same reference as above: BatchNorm over the concatenated batch."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_syncbn_worker, args=(2, port, out, True, use_sinks), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


# ------------------------------------------------------------------------------------------------ fused-accumulation bucket protocol, two processes
class _FusedLinear(torch.autograd.Function):
    """What lavt_hip.ops._Linear does to its weight gradient under the step harness, with the GEMMs in torch: the gradient is written straight into
    the sink (a view of GradBuckets.flat), the Function returns None for it, and -- for `defer` -- the write happens later, at the grouped launch."""

    @staticmethod
    def forward(ctx, x, w, defer, queue, log=None):
        ctx.save_for_backward(x, w)
        ctx.defer, ctx.queue, ctx.log = defer, queue, log
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from lavt_hip import ops
        x, w = ctx.saved_tensors
        wbuf, is_sink = ops.sinks.buf(w, tuple(w.shape))
        assert is_sink
        if ctx.log is not None:
            ctx.log[0].append(("op", ctx.log[1]))
        if ctx.defer:
            ctx.queue.append((wbuf, dy.t() @ x))        # the grouped kernel does not exist yet: PyTorch will still fire w's hook right now
            ops.wgrads.notify(w)
            dw = None
            if ctx.log is not None and len(ctx.queue) == 2:
                ctx.log[2]()                            # the group is full: launch it from inside backward, as _WgradQueue.add does
        else:
            wbuf.copy_(dy.t() @ x)
            dw = ops.sinks.done(w, wbuf, True)
        return dy @ w, dw, None, None, None


class _DeferredAffine(torch.autograd.Function):
    """What the LayerNorm ops do to gamma / beta under the step harness: the partial sums are parked, the parameters are `pending`
    (lavt_hip.ops.ln_deferred.add) and ONE reduction at the end of backward writes the gradients and reports them."""

    @staticmethod
    def forward(ctx, x, g, b, parked):
        ctx.save_for_backward(x, g)
        ctx.params, ctx.parked = (g, b), parked
        return x * g + b

    @staticmethod
    def backward(ctx, dy):
        from lavt_hip import ops
        x, g = ctx.saved_tensors
        for prm, val in zip(ctx.params, ((dy * x).sum(0), dy.sum(0))):
            buf, is_sink = ops.sinks.buf(prm, tuple(prm.shape))
            assert is_sink
            ctx.parked.append((buf, val))
            ops.ln_deferred.params.append(prm)
            ops.wgrads.pending.add(id(prm))
        return dy * g, None, None, None


def _fused_worker(rank, world, port, bucket_mib, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavt_hip import ops
    from lavt_hip.ddp import GradBuckets
    torch.manual_seed(0)
    ws = nn.ParameterList([nn.Parameter(torch.randn(8, 8) * 0.3) for _ in range(6)])
    unused = nn.Parameter(torch.randn(8, 8))
    holder = nn.Module()
    holder.ws0 = nn.ParameterList(list(ws[:3]))
    holder.norm = nn.LayerNorm(8)                   # its gamma / beta stand for the deferred class: registered in the MIDDLE of the network
    holder.unused = unused
    holder.ws1 = nn.ParameterList(list(ws[3:]))
    with torch.no_grad():
        holder.norm.weight.add_(0.1 * torch.randn(8))
        holder.norm.bias.add_(0.1 * torch.randn(8))
    gb = GradBuckets(holder, bucket_mib=bucket_mib, fused_accumulation=True)
    assert gb.late_bucket is not None and gb.bucket_of[holder.norm.weight] == gb.late_bucket == gb.bucket_of[holder.norm.bias]
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(20 + r)) for r in range(world)]
    events = []
    orig_launch = gb._launch

    def logged_launch(b):
        events.append(("launch", b))
        orig_launch(b)
    gb._launch = logged_launch
    try:
        for step in range(2):
            gb.zero()
            del events[:]
            queue, parked = [], []

            def flush():                         # the grouped launch: writes the queued gradients, THEN reports their parameters ready
                for buf, val in queue:
                    buf.copy_(val)
                queue.clear()
                ops.wgrads.flush()
            h = xs[rank]
            for i, w in enumerate(ws):
                h = torch.tanh(_FusedLinear.apply(h, w, i % 2 == 1, queue, (events, i, flush)))     # odd layers: weight gradient deferred to a grouped launch
                if i == 2:
                    h = _DeferredAffine.apply(h, holder.norm.weight, holder.norm.bias, parked)
            loss = h.pow(2).mean()
            loss.backward()                      # the queue flushes itself inside backward whenever two gradients are queued (the Swin block's grouped launch)
            flush()                              # TrainStep._body: ops.wgrads.flush() after backward
            for buf, val in parked:              # the one deferred reduction launch (ops.ln_deferred.flush() inside finish() reports its parameters)
                buf.copy_(val)
            n_before_finish = len(events)
            gb.finish()
            if step == 1:
                # launch order against op order: every ordinary bucket was launched before finish(), and -- with one parameter per bucket -- the
                # bucket of layer i's weight before the backward op of layer i-3 ran; only the late bucket (norm, unused) waits for finish()
                launches = [(k, b) for k, (what, b) in enumerate(events) if what == "launch"]
                in_finish = [b for k, b in launches if k >= n_before_finish]
                assert in_finish == [gb.late_bucket], (in_finish, gb.late_bucket)
                assert gb.bucket_of[unused] == gb.late_bucket
                assert [w for _, _, w in gb.launch_log if w == "finish"] == ["finish"]
                last_op = max(k for k, (what, i) in enumerate(events) if what == "op")
                early = [b for k, b in launches if k < last_op]
                if len(gb.buckets) > 3:          # tiny buckets: one per weight
                    pos = {b: k for k, b in launches}
                    opk = {i: k for k, (what, i) in enumerate(events) if what == "op"}
                    for i in (5, 4, 3):          # (layer 5's gradient is queued until layer 3's joins it: its bucket goes out inside op 3)
                        assert pos[gb.bucket_of[ws[i]]] < opk[i - 3], f"bucket of layer {i} not launched while backward was running"
                    assert len(early) >= 4
        grads = []
        for r in range(world):
            ref = [w.detach().clone().requires_grad_(True) for w in ws]
            gr, br = holder.norm.weight.detach().clone().requires_grad_(True), holder.norm.bias.detach().clone().requires_grad_(True)
            h = xs[r]
            for i, w in enumerate(ref):
                h = torch.tanh(h @ w.t())
                if i == 2:
                    h = h * gr + br
            h.pow(2).mean().backward()
            grads.append([w.grad for w in ref] + [gr.grad, br.grad])
        for i, w in enumerate(list(ws) + [holder.norm.weight, holder.norm.bias]):
            want = sum(g[i] for g in grads) / world
            assert torch.allclose(w.grad, want, atol=1e-6), f"weight {i} (deferred: {i % 2 == 1})"
        assert float(unused.grad.abs().max()) == 0.0
        out[rank] = True
    finally:
        ops.sinks.clear()
        ops.wgrads.ready, ops.wgrads.pending = [], set()
        ops.ln_deferred.params = []
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mib", [64.0, 0.0002])
def test_fused_accumulation_buckets_world2(bucket_mib):
    """TrainStep's gradient protocol between two processes: gradients written into the flat buffer by the ops themselves (no AccumulateGrad),
    half of them deferred to a later 'grouped launch' while PyTorch fires their hooks early, LayerNorm-class gradients reduced by one deferred
    launch at the end, a parameter that never gets a gradient -- buckets all-reduced when complete, the launch order recorded against the op
    order (only the late bucket waits for finish()), and after finish() every .grad is the mean over the ranks."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fused_worker, args=(2, port, bucket_mib, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}
