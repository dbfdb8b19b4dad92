"""Data-parallel gradient exchange for one process per GPU: bucketed all-reduce (RCCL on GPUs, `nccl` backend)
overlapped with backward.

Replaces `torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` of the reference
(train.py:591-592) for the step harness:

* every parameter's `.grad` is a view into one flat fp32 buffer (no per-step flatten/copy);
* buckets are filled in reverse registration order (~ the order autograd produces gradients); the moment the
  last gradient of a bucket has been accumulated, its slice is all-reduced (ReduceOp.AVG) on a side stream while
  backward continues on the compute stream;
* parameters that never receive a gradient (`backbone.layers.3.res_gate.*`) simply keep zeros and
  are reduced with whatever bucket they sit in at `finish()`: no unused-parameter graph traversal;
* no per-forward buffer broadcast: `relative_position_index` is a constant and BatchNorm running statistics are
  identical on every rank by construction (SyncBN semantics in lavt_hip.ops.batch_norm_relu).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 475 MB fp32 (Swin-B) is ~5 ms, comparable
to the backward itself at batch 2/GPU, hence large buckets (default 64 MiB: few, large collectives) and overlap.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

# LAVT_FORCE_COLLECTIVES=1: issue the collectives even in a 1-rank group (exercises the RCCL / stream plumbing on a 1-GPU box)
FORCE_COLLECTIVES = os.environ.get("LAVT_FORCE_COLLECTIVES", "0") == "1"


class GradBuckets:
    def __init__(self, module: torch.nn.Module, bucket_mib: float = 64.0, group=None, broadcast_params: bool = True, fused_accumulation: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # reverse registration order ~ gradient production order
        order = list(reversed(self.params))
        cap = int(bucket_mib * (1 << 20) / 4)
        self.buckets = []          # (start, end) element ranges of `flat`
        self.bucket_of = {}
        off = 0
        cur_start, cur_n = 0, 0
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.bucket_of[p] = len(self.buckets)
            off += n
            cur_n += n
            if cur_n >= cap:
                self.buckets.append((cur_start, off))
                cur_start, cur_n = off, 0
        if cur_n > 0:
            self.buckets.append((cur_start, off))
        self.pending = [0] * len(self.buckets)
        self.expected = [0] * len(self.buckets)
        for p in order:
            self.expected[self.bucket_of[p]] += 1
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._seen = set()
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.fused = fused_accumulation
        # LAVT_BF16_BUCKETS=1: the buckets travel as bf16 (half the bytes per step over xGMI: 238 instead of 475 MB for Swin-B); the flat fp32 buffer
        # the optimizer reads is kept -- a bucket is cast, reduced and cast back on the communication stream (two element-wise passes per bucket)
        self.bf16 = os.environ.get("LAVT_BF16_BUCKETS", "0") == "1"
        self._tmp = []
        if fused_accumulation:          # weight-gradient kernels accumulate straight into `flat` (lavt_hip.ops.sinks)
            from . import ops
            ops.sinks.set(self.params, on_ready=self._on_grad)
        if broadcast_params and self.world > 1:
            for p in module.parameters():
                dist.broadcast(p.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            for b in module.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

    # ---- step protocol: zero() -> forward/backward -> finish() -------------------------------------------------
    def zero(self):
        self.flat.zero_()
        for p in self.params:                      # an optimizer / user may have detached .grad; re-point it
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 0 and not self._is_view(p):
                self._repoint(p)
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._seen = set()
        if self.fused:
            from . import ops
            ops.sinks.begin_step()

    def _is_view(self, p):
        lo = self.flat.data_ptr()
        return p.grad is not None and lo <= p.grad.data_ptr() < lo + self.flat.numel() * 4

    def _repoint(self, p):
        off = 0
        for q in reversed(self.params):
            if q is p:
                break
            off += q.numel()
        p.grad = self.flat[off:off + p.numel()].view_as(p)

    def _on_grad(self, p):
        # A parameter can report twice in one backward: once from the fused-accumulation path (ops.sinks.done, right after its
        # weight-gradient kernel is enqueued) and once from autograd's post-accumulate hook, which PyTorch also runs when the
        # op returned no gradient tensor.  Count each parameter once per step, or a bucket is reduced before it is complete.
        if self.fused:
            from . import ops
            if id(p) in ops.wgrads.pending:          # hook fired for a weight gradient that is still queued for a grouped launch: not ready yet
                return
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.bucket_of[p]
        self.pending[b] += 1
        if self.pending[b] == self.expected[b] and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        self.launched[b] = True
        if self.world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
            return
        s, e = self.buckets[b]
        chunk = self.flat[s:e]
        mode = os.environ.get("LAVT_DDP_MODE", "async_side")
        if self.comm_stream is not None and mode == "sync_main":
            dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
        elif self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if self.fused:
                from . import ops
                for lst in ops.side.streams.values():       # fused wgrad kernels write the bucket from side streams
                    for st in lst:
                        self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                if self.bf16:
                    half = chunk.to(torch.bfloat16)
                    dist.all_reduce(half, op=dist.ReduceOp.AVG, group=self.group)          # enqueued on the communication stream: stream-ordered with the casts
                    chunk.copy_(half)
                    self._tmp.append(half)
                elif mode == "sync_side":
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        elif self.bf16:                                 # gloo (CPU tests): no AVG op; same rounding points as the GPU form
            half = chunk.to(torch.bfloat16)
            w = dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.works.append((w, chunk, half))
        else:                                           # gloo (CPU tests): no AVG op
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.works.append((w, chunk))

    def finish(self):
        """Call after backward: reduces buckets that never filled (unused parameters) and joins the side stream."""
        if self.fused:                                   # gradients still parked by the ops (grouped weight gradients, deferred LayerNorm sums)
            from . import ops
            ops.wgrads.flush()
            ops.ln_deferred.flush()
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for w in self.works:
            if isinstance(w, tuple):
                w[0].wait()
                if len(w) == 3:
                    w[1].copy_(w[2].float() / self.world)
                else:
                    w[1].div_(self.world)
            else:
                w.wait()
        self._tmp = []
        if self.comm_stream is not None and (self.world > 1 or FORCE_COLLECTIVES):
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.works = []

    def bytes_per_step(self) -> int:
        return self.flat.numel() * 4
