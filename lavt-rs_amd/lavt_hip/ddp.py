"""Data-parallel gradient exchange for one process per GPU: bucketed all-reduce (RCCL on GPUs, `nccl` backend)
overlapped with backward.

Replaces `torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` of the reference
(train.py:591-592) for the step harness:

* every parameter's `.grad` is a view into one flat fp32 buffer (no per-step flatten/copy);
* buckets are filled in reverse registration order (~ the order autograd produces gradients); the moment the
  last gradient of a bucket has been accumulated, its slice is all-reduced (ReduceOp.AVG) on a side stream while
  backward continues on the compute stream;
* parameters whose gradient exists only when backward has ended -- LayerNorm / bias-table sums that the step harness reduces with one
  deferred launch, and parameters that never receive a gradient (`backbone.layers.3.res_gate.*`, SURVEY.md B6: they keep zeros) -- sit in
  ONE late bucket at the end of the flat buffer, reduced at `finish()` (1.03 MB + 8.4 MB of Swin-B's 475 MB); every other bucket is
  reduced while backward is still running.  No unused-parameter graph traversal: a parameter that did not report in the first step
  joins the late bucket from the second step on;
* no per-forward buffer broadcast: `relative_position_index` is a constant and BatchNorm running statistics are
  identical on every rank by construction (SyncBN semantics in lavt_hip.ops.batch_norm_relu).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 475 MB fp32 (Swin-B) is ~5 ms, comparable
to the backward itself at batch 2/GPU, hence large buckets (default 32 MiB, `engine.TrainStep(bucket_mib=...)` / LAVT_BUCKET_MIB: few, large collectives) and overlap.
"""
import os
import sys
from typing import List, Optional

import torch
import torch.distributed as dist

# LAVT_FORCE_COLLECTIVES=1: issue the collectives even in a 1-rank group (exercises the RCCL / stream plumbing on a 1-GPU box)
FORCE_COLLECTIVES = os.environ.get("LAVT_FORCE_COLLECTIVES", "0") == "1"


def late_gradient_parameters(module: torch.nn.Module):
    """Parameters whose gradient only exists once backward has ENDED, whatever position their layer has in the network:

    * LayerNorm weight / bias and `relative_position_bias_table`: the step harness parks their per-workgroup partial sums and reduces all of them
      with one launch at the end of backward (lavt_hip.ops.ln_deferred; 258 296 floats = 1.03 MB for Swin-B);
    * parameters that never receive a gradient (`module.never_used_parameters()` when the model says so: `backbone.layers.3.res_gate.*`,
      SURVEY.md B6 -- the reference needs find_unused_parameters=True for them): they stay zero and nothing ever reports them ready.

    Mixed into the ordinary buckets they gate those buckets until `finish()` (round 3: 393 of Swin-B's 475 MB waited for 1 MB of such parameters);
    GradBuckets gives them a bucket of their own that is reduced last."""
    late = {}
    for m in module.modules():
        if isinstance(m, torch.nn.LayerNorm):
            for p in m.parameters(recurse=False):
                late[id(p)] = p
    for n, p in module.named_parameters():
        if n.endswith("relative_position_bias_table"):
            late[id(p)] = p
    for m in module.modules():
        fn = getattr(m, "never_used_parameters", None)
        if callable(fn):
            for p in fn():
                late[id(p)] = p
    return [p for p in late.values() if p.requires_grad]


# Moves every time a GradBuckets instance (re-)lays out its flat buffer: every p.grad address may change then (first construction, and the relayout in
# the second zero() that sends unreported parameters to the late bucket).  FusedAdamW folds it into its descriptor-table key, so that
# step(check_tables=False) -- which skips the host-side pointer scan -- refuses a table built for the previous layout instead of reading stale offsets.
layout_generation = [0]


class GradBuckets:
    def __init__(self, module: torch.nn.Module, bucket_mib: float = 64.0, group=None, broadcast_params: bool = True, fused_accumulation: bool = False,
                 late="auto"):
        """late: "auto" = late_gradient_parameters(module); a list of parameters; or None / [] for one reverse-registration-order sequence of buckets"""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.cap = int(bucket_mib * (1 << 20) / 4)
        self._names = {id(p): n for n, p in module.named_parameters()}
        late_ids = {id(p) for p in (late_gradient_parameters(module) if isinstance(late, str) else (late or []))}
        self._layout(late_ids)
        self.works = []
        self._seen = set()
        self._relayout = None       # parameters of ordinary buckets that had not reported when finish() was called (first step): moved to the late bucket
        self._steps = 0
        self.launch_log = []        # per step: (bucket, number of parameters that had reported, "backward" | "finish")
        self._in_finish = False
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.fused = fused_accumulation
        # LAVT_BF16_BUCKETS=1: the buckets travel as bf16 (half the bytes per step over xGMI: 238 instead of 475 MB for Swin-B); the flat fp32 buffer
        # the optimizer reads is kept -- a bucket is cast, reduced and cast back on the communication stream (two element-wise passes per bucket)
        self.bf16 = os.environ.get("LAVT_BF16_BUCKETS", "0") == "1"
        self.mode = os.environ.get("LAVT_DDP_MODE", "async_side")
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

    def _layout(self, late_ids):
        """flat = [ordinary parameters in reverse registration order ~ the order backward produces their gradients, cut into buckets of `cap`
        floats] + [the late parameters: ONE bucket, reduced at finish()].  The layout depends on the module alone: identical on every rank."""
        self.late_ids = set(late_ids)
        layout_generation[0] += 1
        self._zero_views, self._skip_views = None, []          # (a zero-fill skip list refers to the old offsets)
        order = [p for p in reversed(self.params) if id(p) not in self.late_ids]
        late = [p for p in reversed(self.params) if id(p) in self.late_ids]
        self.buckets = []          # (start, end) element ranges of `flat`
        self.bucket_of = {}
        self.offset_of = {}
        off = 0
        cur_start, cur_n = 0, 0
        for p in order:
            n = p.numel()
            self.offset_of[id(p)] = off
            p.grad = self.flat[off:off + n].view_as(p)
            self.bucket_of[p] = len(self.buckets)
            off += n
            cur_n += n
            if cur_n >= self.cap:
                self.buckets.append((cur_start, off))
                cur_start, cur_n = off, 0
        if cur_n > 0:
            self.buckets.append((cur_start, off))
        self.late_bucket = None
        if late:
            self.late_bucket = len(self.buckets)
            start = off
            for p in late:
                n = p.numel()
                self.offset_of[id(p)] = off
                p.grad = self.flat[off:off + n].view_as(p)
                self.bucket_of[p] = self.late_bucket
                off += n
            self.buckets.append((start, off))
        self.expected = [0] * len(self.buckets)
        for p in self.params:
            self.expected[self.bucket_of[p]] += 1
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)

    def set_zero_skip(self, param_ids):
        """Parameters whose weight-gradient launch overwrites the whole gradient with plain stores (ops.sinks.assigned after a step: the grouped Linear
        gradients, the fused-tap convolution gradients -- 102 of Swin-B LAVT's 119 M values) need no zero fill: zero() then clears only the rest of the
        flat buffer, as one multi-tensor launch (475 MB at the HBM write rate were 58 us at the head of every step).  Only the step harness sets this,
        for a CAPTURED step whose launch sequence is frozen, and it verifies the set on the captured graph (engine.TrainStep); None / empty = fill everything."""
        self._zero_views, self._skip_views = None, []
        ids = {i for i in (param_ids or ()) if i in self.offset_of}
        if not ids:
            return 0
        spans = sorted((self.offset_of[id(p)], p.numel()) for p in self.params if id(p) in ids)
        views, pos = [], 0
        for off, n in spans:
            if off > pos:
                views.append(self.flat[pos:off])
            pos = max(pos, off + n)
        if pos < self.flat.numel():
            views.append(self.flat[pos:])
        self._zero_views = views
        self._skip_views = [self.flat[off:off + n] for off, n in spans]
        return sum(n for _, n in spans)

    def overlappable_bytes(self) -> int:
        """bytes of the buckets that can be reduced while backward is still running (every bucket but the late one)"""
        return sum(4 * (e - s) for b, (s, e) in enumerate(self.buckets) if b != self.late_bucket)

    # ---- step protocol: zero() -> forward/backward -> finish() -------------------------------------------------
    def zero(self, defer_fill=False, also_zero=None):
        """defer_fill (step harness): the zero fill of the flat buffer is handed to lavt_hip.ops.fill_riders -- forward launches zero it slice by slice
        with rider workgroups, and the harness calls ops.fill_riders.finish() before backward starts"""
        if self._relayout:          # learnt in the first step: parameters nothing reports during backward join the late bucket
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GradBuckets: the bucket layout changes after the first step; run one eager step before capturing")
            self._layout(self.late_ids | self._relayout)
            if self.fused:
                from . import ops
                ops.sinks.set(self.params, on_ready=self._on_grad)
        self._relayout = None
        taken = False
        if defer_fill:
            from . import ops
            taken = ops.fill_riders.begin(self.flat)
        if not taken:
            if self._zero_views is not None:
                # everything but the parameters whose gradient launch overwrites its buffer (set_zero_skip); `also_zero`: a float32 tensor of the caller's
                # (the step harness's arena of small zero-initialised buffers) cleared by the same multi-tensor launch
                torch._foreach_zero_(self._zero_views + [also_zero] if also_zero is not None else self._zero_views)
                also_zero = None
            else:
                self.flat.zero_()
        if also_zero is not None:
            also_zero.zero_()
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + self.flat.numel() * 4
        for p in self.params:                      # an optimizer / user may have detached .grad; re-point it
            if p.grad is None or p.grad.data_ptr() != lo + 4 * self.offset_of[id(p)]:
                o = self.offset_of[id(p)]
                p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.pending = [0] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._seen = set()
        self.launch_log = []
        self._in_finish = False
        if self.fused:
            from . import ops
            ops.sinks.begin_step()

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
        self.launch_log.append((b, len(self._seen), "finish" if self._in_finish else "backward"))
        if self.world == 1 and not (FORCE_COLLECTIVES and dist.is_initialized()):
            return
        s, e = self.buckets[b]
        chunk = self.flat[s:e]
        mode = self.mode
        if self.comm_stream is not None and mode == "sync_main":
            if self.bf16:
                half = chunk.to(torch.bfloat16)
                dist.all_reduce(half, op=dist.ReduceOp.AVG, group=self.group)
                chunk.copy_(half)
            else:
                dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
        elif self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                if self.bf16:                           # (always stream-ordered on the communication stream, whatever LAVT_DDP_MODE says)
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
        """Call after backward: reduces the late bucket (deferred LayerNorm / bias-table sums, never-used parameters) and joins the side stream."""
        self._in_finish = True
        if self.fused:                                   # weight gradients still queued for a grouped launch: the tail of backward itself
            from . import ops
            ops.wgrads.flush()
        if self._steps == 0:
            # First step: an ordinary bucket that is still incomplete now would be reduced only here, every step -- after backward, not beside it.
            # Whatever has not reported by now (a parameter the forward never uses, a deferred reduction late_gradient_parameters() did not
            # know about) moves to the late bucket from the next step on.  Deterministic given the model: the same decision on every rank.
            miss = {id(p) for p in self.params if id(p) not in self._seen and id(p) not in self.late_ids}
            if self.world > 1 and dist.is_initialized():
                # the new layout must be the same on every rank (bucket boundaries = all-reduce chunk sizes): the union of the ranks' unreported
                # sets decides, so a parameter whose use was data-dependent in step 1 cannot give two ranks two layouts
                mask = torch.tensor([1 if id(p) in miss else 0 for p in self.params], dtype=torch.int32,
                                    device=self.flat.device if dist.get_backend(self.group) == "nccl" else "cpu")
                dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
                miss = {id(p) for p, f in zip(self.params, mask.tolist()) if f and id(p) not in self.late_ids}
            if miss:
                self._relayout = miss
                if os.environ.get("LAVT_DDP_VERBOSE", "0") == "1":
                    print("[lavt_hip.ddp] no gradient during backward, moved to the late bucket: " + ", ".join(sorted(self._names.get(i, "?") for i in miss)), file=sys.stderr)
        self._steps += 1
        if self.fused:                                   # deferred LayerNorm / bias-table sums: one reduction launch, reports the late bucket's members
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
