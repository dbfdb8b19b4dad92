"""Step harness reproducing the reference caller's sequence (train.py:199-229) on the HIP model:

    zero grads -> out = model(image, l, l_mask) -> F.cross_entropy(out, target, weight=[0.9, 1.1]) -> backward
    -> gradient all-reduce (world > 1)

Gradient accumulation over micro-batches is not supported by this harness (every parameter receives exactly one weight gradient per
step into a buffer zeroed at the start of the step; lavt_hip.ops.sinks refuses a second one).

The whole step is ~1,000 kernel launches for Swin-B; issued from Python it is launch-bound, so on one GPU the
step is captured once into a hipGraph (torch.cuda.CUDAGraph on the stream our C-ABI launches go to) and
replayed: weight casts, DropPath masks, BatchNorm running-stat updates all happen inside the graph.
With world > 1 the same capture includes the collectives: the SyncBN statistic all-reduces on the capture stream and the
bucketed gradient all-reduces on the communication stream (forked from / joined to the capture stream with events), so a
replay issues forward, backward and the overlapped RCCL all-reduces without any Python in between.  RCCL supports stream
capture (its collectives become graph kernel nodes); `tools/nccl_graph_probe.py` and tests/test_gpu_modules.py exercise the
mechanics on one GPU in a 1-rank group.  LAVT_DDP_GRAPH=0 (or a failed capture) falls back to eager launching, where the
step is bound by the ~23 ms of host-side launch work.
"""
import os
import sys

import torch
import torch.nn.functional as F

from . import ops
from .ddp import GradBuckets
from .runtime import compute_dtype, fp8_enabled


def _in_context(fn):
    """run a TrainStep method inside the step's own ops.StepContext"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with ops.use_context(self.context):
            return fn(self, *a, **kw)
    return wrapped


class TrainStep:
    def __init__(self, model, image, l_feats, l_mask, target, world=1, use_graph=True, bucket_mib=32.0, fused_loss=True, refresh_weights_in_step=False, context=None):
        """context: the ops.StepContext this harness keeps its state in (gradient sinks, deferred-launch queues, weight copies, scratch).  None = the
        process-wide default context -- what the drop-in path and a single harness use.  Give every further model in the process its own
        `ops.StepContext()` (and its optimizer the same one: FusedAdamW(..., context=)): their steps can then alternate freely."""
        self.context = context if context is not None else ops.default_context()
        with ops.use_context(self.context):
            self._init(model, image, l_feats, l_mask, target, world, use_graph, bucket_mib, fused_loss, refresh_weights_in_step)

    def _init(self, model, image, l_feats, l_mask, target, world, use_graph, bucket_mib, fused_loss, refresh_weights_in_step):
        self.model = model
        dev = image.device
        self.x, self.l, self.m, self.t = image, l_feats, l_mask, target
        self.w = torch.tensor([0.9, 1.1], device=dev)                    # losses.py:7-11
        # 32 MiB: the bucket that holds the earliest layers is reduced after backward has ended -- its all-reduce is the exposed tail of the step
        # (64 MiB ~ 0.4-0.8 ms over xGMI), while ~15 collectives of this size still run at full ring bandwidth.  LAVT_BUCKET_MIB overrides.
        bucket_mib = float(os.environ.get("LAVT_BUCKET_MIB", bucket_mib))
        self.buckets = GradBuckets(model, bucket_mib=bucket_mib, fused_accumulation=True)
        self.world = world
        self.graph = None
        self.loss = None
        self.use_graph = use_graph and (world == 1 or os.environ.get("LAVT_DDP_GRAPH", "1") != "0")
        self.captured = False
        self.zero_skip_values = 0                # gradient values left out of the captured zero fill (GradBuckets.set_zero_skip)
        ops.wgrads.enabled = True                # one weight gradient per parameter per step into the zeroed flat buffer: grouped, plainly stored
        self.fused_loss = fused_loss and hasattr(model, "forward_lowres")
        self.stats = None                        # fused loss: [loss, sum of weights, I, U] of the last step (device tensor)
        # The compute-dtype weight copies are refreshed by the optimizer (FusedAdamW.step re-casts them right after the update).  Set this when
        # the weights are changed by something else between replays of the captured step (a torch.optim optimizer, manual edits): the casts
        # (~0.3 ms for Swin-B) then run at the start of every step, inside the graph.
        self.refresh_in_step = refresh_weights_in_step or os.environ.get("LAVT_REFRESH_IN_STEP", "0") == "1"
        # A captured step runs no Python, so the per-parameter version check of the weight cache never fires on replay.  step() therefore compares
        # the parameters' version counters / storage addresses with what the compute copies were made from and re-casts them (eagerly, in front of
        # the replay) when anything but FusedAdamW has touched them: torch.optim optimizers, load_state_dict, manual edits all bump p._version.
        # FusedAdamW writes through raw pointers (versions do not move) and refreshes the copies itself.
        self._params = [p for p in model.parameters()]
        self._seen = None
        self._one = None

    def _param_stamp(self):
        return sum(p._version for p in self._params), sum(p.data_ptr() for p in self._params)

    @_in_context
    def _body(self):
        if self.refresh_in_step:
            ops.weights.refresh_all()            # re-cast weights inside the step (for optimizers that do not maintain the compute copies)
        arena = ops.zero_arena.begin_step(self.x.device, defer=True)          # one fill for every small zero-initialised buffer of the step ...
        self.buckets.zero(defer_fill=True, also_zero=arena)                    # ... shared with the gradient buffer's (the 475 MB fill of an unskipped buffer rides on forward launches: ops.fill_riders)
        ops.dtable_chain.job, ops.dtable_chain.keep = None, None          # (a backward that raised mid-way must not leave its binning job to the next step)
        if fp8_enabled():
            ops.fp8.advance()                    # delayed scaling: last step's |max| values become this step's quantisation scales
        if self.fused_loss:                       # upsample + weighted CE (+ I/U) fused: the (B,2,H,W) logits are never written
            from lib._utils import fused_loss
            loss, self.stats = fused_loss(self.model.forward_lowres(self.x, self.l, self.m), self.t, (0.9, 1.1))
        else:
            out = self.model(self.x, self.l, self.m)
            loss = F.cross_entropy(out, self.t, weight=self.w)
        ops.fill_riders.finish()                 # whatever of the gradient buffer no forward launch has zeroed
        if self._one is None or self._one.shape != loss.shape or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)    # (first eager step) the root gradient as a persistent tensor: `loss.backward()` fills a fresh ones_like every step,
        loss.backward(self._one)                 # one more 4.5 us launch on the captured chain
        ops.wgrads.flush()                       # weight-gradient GEMMs still queued for a grouped launch
        ops.ln_deferred.flush()                  # all LayerNorm weight / bias partial sums of this backward: one reduction launch
        self.buckets.finish()                    # stragglers (never-used parameters) + join of the communication stream
        ops.zero_arena.end_step()
        ops.fp8.end_step()
        return loss.detach()

    @_in_context
    def warmup_and_capture(self, eager_iters=3):
        """Side effects beyond `eager_iters` steps: one more eager step when the bucket layout is still to settle (eager_iters = 1), and the zero-fill-skip validation below replays the captured step up to three more times on
        NaN-poisoned gradient buffers.  Each of those replays is a real training step of the forward pass -- BatchNorm running statistics and
        `num_batches_tracked`, the fp8 |max| history and the DropPath generator advance, and with world > 1 the poisoned buckets are all-reduced
        (NaN on every rank alike) -- the gradients of such a replay are discarded by the next step's fill.  LAVT_ZERO_SKIP=0 captures once."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for it in range(eager_iters):
                self.loss = self._body()
                if it == 0:                      # every compute copy exists now: one descriptor table for the one-launch refresh
                    ops.weights.build_multicast(compute_dtype())
                    ops.weights.refresh_all()
            if self.use_graph and self.buckets._relayout:
                # GradBuckets lays the flat buffer out again in the zero() after its first step (parameters nothing reported join the late bucket): that
                # moves p.grad and must happen in an eager step -- inside the capture window zero() would raise and the harness would silently run eagerly
                self.loss = self._body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._seen = self._param_stamp()
        if not self.use_graph:
            return
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # ProcessGroupNCCL's watchdog thread retires the eager warm-up collectives by polling their events every ~100 ms.  Give it time to
            # empty its list before the capture window opens: a poll that lands inside the window was seen (1 run in ~10 of the 1-rank-group
            # test) to fail with hipErrorCapturedEvent and take the process down, also under the thread_local capture mode used below.
            # Explicit drain first: every warm-up collective has been waited for (GradBuckets.finish), the device is idle (synchronize above), and a
            # barrier puts all ranks at the same point -- after it no rank has an incomplete collective.  What remains is the watchdog's own
            # bookkeeping: it drops completed work objects from its list on its next poll, and PyTorch exposes no call to wait for that, so the
            # harness waits a few poll periods (TORCH_NCCL watchdog sleep: 100 ms) before opening the capture window.
            import time
            torch.distributed.barrier()
            torch.cuda.synchronize()
            time.sleep(float(os.environ.get("LAVT_CAPTURE_SETTLE_S", "0.35")))
        # Zero-fill skip: parameters whose gradient launch overwrote its buffer with plain stores in the last eager step (ops.sinks.assigned: members of the
        # grouped weight-gradient launches, the fused-tap convolution gradients) are CANDIDATES for being left out of the captured zero fill.  The captured
        # graph itself decides: the candidates' gradients are poisoned, the graph replayed once, and whatever still holds NaN -- a member the library cut
        # into pieces that meet through atomics, a buffer not fully written -- goes back into the fill and the step is captured again (at most three
        # rounds; the launch sequence of a captured step is frozen, so what passes here holds for every replay).
        cand = set(ops.sinks.assigned) if os.environ.get("LAVT_ZERO_SKIP", "1") != "0" else set()
        self.zero_skip_values = 0
        for attempt in range(4):
            skipped = self.buckets.set_zero_skip(cand if attempt < 3 else None)
            self._capture()
            if not self.captured:
                self.buckets.set_zero_skip(None)         # eager fallback: the full fill
                break
            if not skipped:
                break
            import math
            for v in self.buckets._skip_views:
                v.fill_(math.nan)
            self.graph.replay()
            torch.cuda.synchronize()
            flags = torch.stack([v.isnan().any() for v in self.buckets._skip_views]).tolist()
            if not any(flags):
                self.zero_skip_values = skipped
                break
            by_start = {self.buckets.flat.data_ptr() + 4 * self.buckets.offset_of[id(p)]: p for p in self.buckets.params}
            bad = [by_start[v.data_ptr()] for v, f in zip(self.buckets._skip_views, flags) if f]
            cand -= {id(p) for p in bad}
            if os.environ.get("LAVT_ZERO_SKIP_VERBOSE"):
                names = {id(p): n for n, p in self.model.named_parameters()}
                print(f"[lavt_hip.engine] zero-fill skip: {len(bad)} candidates are accumulated into or not fully written (e.g. {', '.join(names.get(id(p), '?') for p in bad[:4])}); "
                      "they stay in the fill, capturing again", file=sys.stderr)
            self.graph = None

    def _capture(self):
        self.captured = False
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: ProcessGroupNCCL's watchdog thread polls events of earlier (eager warm-up) collectives with hipEventQuery; under the
            # default "global" capture mode such a call from another thread, if it lands inside the capture window, aborts the capture with
            # hipErrorStreamCaptureUnsupported (seen as a c10::DistBackendError that takes the process down)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.loss = self._body()
            self.graph = g
            self.captured = True
        except Exception as e:                                   # noqa: BLE001  (report and keep the eager path)
            print(f"[lavt_hip.engine] hipGraph capture failed, running eagerly: {type(e).__name__}: {e}", file=sys.stderr)
            self.graph = None
            torch.cuda.synchronize()

    @_in_context
    def step(self):
        if not self.refresh_in_step:
            stamp = self._param_stamp()
            if stamp != self._seen:                 # weights changed behind the compute copies' back (see __init__): refresh before the step
                if self._seen is not None:
                    ops.weights.refresh_all()
                self._seen = stamp
        if self.graph is not None:
            self.graph.replay()
        else:
            self.loss = self._body()
        return self.loss
