"""Autograd-aware host wrappers over the liblavt_hip C ABI.

Each torch.autograd.Function below launches hand-written gfx950 kernels through ctypes on the
current HIP stream, with raw device pointers taken from torch tensors (torch is the allocator and
stream provider only).  There is no alternative implementation: CPU tensors raise.

Activations are 2-D [rows, channels] tensors in the compute dtype (float32 = exact parity path,
bfloat16 = MFMA throughput path).  Parameters stay fp32 nn.Parameters (reference state-dict);
bf16 compute copies are produced by `weight()` and cached per parameter version.  All parameter
gradients are produced in fp32.
"""
import ctypes as C
import os
import types
import weakref
from dataclasses import dataclass
from typing import Optional

import torch

from . import _capi as K
from .runtime import fp8_enabled

KV_LD = 32          # padded word axis of PWAM keys/values (N_l <= 32)


# ------------------------------------------------------------------------------------------ step context
# Everything the ops keep BETWEEN calls -- compute-dtype weight copies, gradient sinks, the queues of deferred launches, zero arenas, fp8 scale history,
# per-device scratch, the DropPath generator -- lives in ONE StepContext (defined below, after the classes it bundles).  The module-level names
# (`weights`, `sinks`, `wgrads`, ...) are proxies onto the CURRENT context: the drop-in path (the reference's train.py / test.py loop) runs in the default
# context and never sees the difference; a harness that owns a private context (engine.TrainStep(context=StepContext())) runs its forward, backward and
# optimizer inside `with use_context(ctx):`, so two models -- or a train + eval pair -- in one process cannot share sinks, queues or scratch.
_ctx = None


class _Proxy:
    """module-level stand-in for one member of the current StepContext: attribute reads / writes go to `getattr(_ctx, name)`"""

    def __init__(self, name=None):
        if name is None:                          # `ops.fp8.__init__()` in a test: re-initialise the member it stands for
            getattr(_ctx, object.__getattribute__(self, "_n")).__init__()
            return
        object.__setattr__(self, "_n", name)

    def __getattr__(self, a):
        return getattr(getattr(_ctx, object.__getattribute__(self, "_n")), a)

    def __setattr__(self, a, v):
        setattr(getattr(_ctx, object.__getattribute__(self, "_n")), a, v)


# ------------------------------------------------------------------------------------------ weights
class _WeightCache:
    """compute-dtype copies of fp32 parameters: 'lin' = [N,K] matrix view, 'conv3' = [Cout][9][Cin]."""

    def __init__(self):
        self.store = {}
        self.epoch = 0
        self.generation = 0          # bumped whenever a copy gets a NEW buffer (first use, shape change, re-homing into a stacked buffer): whoever
                                     # baked copy addresses into a device table (FusedAdamW) keys the table on it

    def invalidate(self):
        self.epoch += 1

    def get(self, p: torch.Tensor, dtype: torch.dtype, kind: str = "lin") -> torch.Tensor:
        if kind == "lin" and dtype == torch.float32:
            return p.detach().reshape(p.shape[0], -1)
        key = (id(p), dtype, kind)
        ent = self.store.get(key)
        if ent is not None and ent[2]() is not p:          # id() reuse after the old parameter died
            ent = None
        stamp = (p._version, p.data_ptr(), self.epoch)
        if ent is not None and ent[0] == stamp:
            return ent[1]
        src = p.detach()
        if kind == "lin":
            shape = (src.shape[0], src.numel() // src.shape[0])
        elif kind == "conv3":                        # [Cout][Cin][taps...] -> [Cout][taps][Cin]
            shape = (src.shape[0], src.numel() // src.shape[0])
        else:
            raise KeyError(kind)
        # keep the same storage across refreshes (static addresses for hipGraph replays)
        if ent is not None and tuple(ent[1].shape) == shape:
            out = ent[1]
        else:
            out = torch.empty(shape, dtype=dtype, device=src.device)
            self.generation += 1
        if kind == "lin":
            K.check(K.lib.lavt_cast(K.F32, K.ptr(src), K.dt(dtype), K.ptr(out), src.numel(), K.stream()))
        else:
            K.check(K.lib.lavt_pack_conv3x3(K.ptr(src), K.dt(dtype), K.ptr(out), src.shape[0], src.shape[1],
                                            src.numel() // (src.shape[0] * src.shape[1]), K.stream()))
        self.store[key] = (stamp, out, weakref.ref(p))
        return out


    def get_fp8(self, p: torch.Tensor, kind: str = "lin"):
        """-> (e4m3 copy as a uint8 tensor, device float [1] holding the |max| it was quantised against).  'lin' = [N, K] rows as stored,
        'conv3' = [Cout][taps][Cin], 'conv3t' = [Cin][taps][Cout] (the data gradient's operand).  Current scaling: |max| is recomputed at every refresh."""
        key = (id(p), "fp8", kind)
        ent = self.store.get(key)
        if ent is not None and ent[2]() is not p:
            ent = None
        stamp = (p._version, p.data_ptr(), self.epoch)
        if ent is not None and ent[0] == stamp:
            return ent[1]
        src = p.detach()
        cout = src.shape[0]
        cin = src.shape[1] if kind in ("conv3", "conv3t") else src.numel() // cout
        taps = src.numel() // (cout * cin)
        shape = (cin, taps * cout) if kind == "conv3t" else (cout, taps * cin)
        q, amax = ent[1] if ent is not None else (torch.empty(shape, dtype=torch.uint8, device=src.device), torch.zeros(1, dtype=torch.float32, device=src.device))
        if kind == "conv3t":
            K.check(K.lib.lavt_fp8_quantize_weight_t(K.ptr(src), K.ptr(q), K.ptr(amax), cout, cin, taps, K.stream()))
        else:
            K.check(K.lib.lavt_fp8_quantize_weight(K.ptr(src), K.ptr(q), K.ptr(amax), cout, cin, taps, K.stream()))
        self.store[key] = (stamp, (q, amax), weakref.ref(p))
        return q, amax

    def get_lnfold(self, w, b, gamma, beta):
        """(Wg bf16 [N, K] = gamma-folded weight, wsum fp32 [N], biasp fp32 [N]) of a Linear that directly follows a LayerNorm (lavt_ln_fold): the
        fused W-MSA kernel contracts RAW rows with Wg and applies the normalisation in its epilogue.  Refreshed like every compute copy (by
        refresh_all / on a version change of any of the four parameters)."""
        key = (id(w), "lnfold", "lin")
        ent = self.store.get(key)
        ps = (w, b, gamma, beta)
        if ent is not None and ent[2]() is not w:
            ent = None
        stamp = (tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps), self.epoch)
        if ent is not None and ent[0] == stamp:
            return ent[1][:3]
        N, Kd = w.shape[0], w.numel() // w.shape[0]
        if ent is not None:
            Wg, wsum, biasp = ent[1][:3]
        else:
            Wg = torch.empty(N, Kd, dtype=torch.bfloat16, device=w.device)
            wsum = torch.empty(N, dtype=torch.float32, device=w.device)
            biasp = torch.empty(N, dtype=torch.float32, device=w.device)
        K.check(K.lib.lavt_ln_fold(K.ptr(w.detach()), K.ptr(gamma.detach()), K.ptr(beta.detach()), K.ptr(b.detach()), K.ptr(Wg), K.ptr(wsum), K.ptr(biasp), N, Kd, K.stream()))
        self.store[key] = (stamp, (Wg, wsum, biasp, [weakref.ref(p) for p in ps]), weakref.ref(w))
        return Wg, wsum, biasp

    def get_cat(self, ps, dtype: torch.dtype) -> torch.Tensor:
        """[sum N_i, K] compute copy of several Linear weights stacked along N (BERT's query / key / value): the per-parameter 'lin' entries
        become row slices of ONE buffer, so the usual refresh (get / refresh_all's multi-cast) keeps the stacked matrix current for free."""
        if dtype == torch.float32:                           # exact path: parameters are used in place, the stack is a copy per call
            return torch.cat([p.detach().reshape(p.shape[0], -1) for p in ps], 0)
        cats = self.__dict__.setdefault("cats", {})
        key = tuple(id(p) for p in ps) + (dtype,)
        ent = cats.get(key)
        if ent is None or any(r() is not p for r, p in zip(ent[1], ps)):
            kd = ps[0].numel() // ps[0].shape[0]
            big = torch.empty(sum(p.shape[0] for p in ps), kd, dtype=dtype, device=ps[0].device)
            off = 0
            for p in ps:
                self.store[(id(p), dtype, "lin")] = (None, big[off:off + p.shape[0]], weakref.ref(p))       # stale stamp: the next get() casts into the slice
                off += p.shape[0]
            ent = (big, [weakref.ref(p) for p in ps])
            cats[key] = ent
            self.generation += 1
            self.multi = None                                # a descriptor table built earlier points at the old per-parameter copies
        for p in ps:
            self.get(p, dtype, "lin")
        return ent[0]

    def get_bias_cat(self, bs) -> torch.Tensor:
        """fp32 biases of several Linear layers stacked into ONE persistent vector (the stacked-weight GEMMs' bias operand): rebuilt when a bias changes
        (version / address / refresh_all's epoch -- the optimizer's refresh), not by a torch.cat inside every step."""
        key = (tuple(id(b) for b in bs), "f32", "bcat")
        ent = self.store.get(key)
        if ent is not None and any(r() is not b for r, b in zip(ent[1][1], bs)):
            ent = None
        stamp = (tuple(b._version for b in bs), tuple(b.data_ptr() for b in bs), self.epoch)
        if ent is not None and ent[0] == stamp:
            return ent[1][0]
        n = sum(b.numel() for b in bs)
        out = ent[1][0] if ent is not None and ent[1][0].numel() == n else torch.empty(n, dtype=torch.float32, device=bs[0].device)
        torch.cat([b.detach().float().reshape(-1) for b in bs], out=out)
        self.store[key] = (stamp, (out, [weakref.ref(b) for b in bs]), weakref.ref(bs[0]))
        return out

    # ---- one-launch refresh of every cached 'lin' copy (the per-parameter casts are ~140 tiny launches per step) ----
    def build_multicast(self, dtype):
        """Call after a warm-up step: collects every live (param, 'lin') entry of `dtype` into a device descriptor table."""
        ents = [(k, e) for k, e in self.store.items() if k[1] == dtype and k[2] == "lin" and e[2]() is not None]
        if not ents:
            self.multi = None
            return
        dev = ents[0][1][1].device
        desc = torch.tensor([[e[2]().data_ptr(), e[1].data_ptr(), e[1].numel()] for _, e in ents], dtype=torch.int64)
        self.multi = (dtype, desc.to(dev), [k for k, _ in ents])

    def refresh_all(self, done=frozenset()):
        """Start of a step / end of an optimizer step: make every compute copy current (1 launch for all Linear weights + 1 per 3x3 conv + 1 for all
        LayerNorm folds).  `done`: {key: address} of 'lin' copies the caller has just written itself (FusedAdamW's update kernel): only their
        stamps move -- provided the copy still lives at the address the caller wrote to (a re-homed or re-allocated copy is cast again)."""
        self.epoch += 1
        multi = getattr(self, "multi", None)
        fresh = set()
        for k in done:
            ent = self.store.get(k)
            if ent is not None and ent[2]() is not None and (not isinstance(done, dict) or ent[1].data_ptr() == done[k]):
                p = ent[2]()
                self.store[k] = ((p._version, p.data_ptr(), self.epoch), ent[1], ent[2])
                fresh.add(k)
        if multi is not None:
            dtype, desc, keys = multi
            if all(k in self.store and self.store[k][2]() is not None for k in keys):      # parameters still alive and cached
                if not all(k in fresh for k in keys):
                    K.check(K.lib.lavt_cast_multi(K.ptr(desc), desc.shape[0], K.dt(dtype), K.stream()))
                for k in keys:
                    st, out, ref = self.store[k]
                    p = ref()
                    self.store[k] = ((p._version, p.data_ptr(), self.epoch), out, ref)
                    fresh.add(k)
            else:
                self.multi = None
        folds = []
        for k, (st, out, ref) in list(self.store.items()):
            p = ref()
            if p is None:
                del self.store[k]
            elif k[1] == "fp8":
                self.get_fp8(p, k[2])
            elif k[2] == "bcat":
                bs = [r() for r in out[1]]
                if all(b is not None for b in bs):
                    self.get_bias_cat(bs)
                else:
                    del self.store[k]
            elif k[1] == "lnfold":
                others = [r() for r in out[3]]
                if all(o is not None for o in others):
                    folds.append((k, others, out))
                else:
                    del self.store[k]
            elif k not in fresh:
                self.get(p, k[1], k[2])
        if folds:
            self._refresh_folds(folds)

    def _refresh_folds(self, folds):
        """all LayerNorm folds in ONE launch (48 per Swin-B step otherwise); the device descriptor table is rebuilt only when the set changes"""
        key = tuple((k, tuple(p.data_ptr() for p in ps)) for k, ps, _ in folds)
        tab = getattr(self, "fold_table", None)
        if tab is None or tab[0] != key:
            rows = [[w.data_ptr(), g.data_ptr(), be.data_ptr(), b.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), w.shape[0], w.numel() // w.shape[0]]
                    for _, (w, b, g, be), out in folds]
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("LayerNorm folds: the set of folded weights changed inside a graph capture")
            tab = self.fold_table = (key, torch.tensor(rows, dtype=torch.int64).to(folds[0][1][0].device))
        K.check(K.lib.lavt_ln_fold_multi(K.ptr(tab[1]), len(folds), K.stream()))
        for k, ps, out in folds:
            self.store[k] = ((tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps), self.epoch), out, self.store[k][2])


weights = _Proxy("weights")


class _Fp8State:
    """Delayed-scaling bookkeeping of the fp8 path: one |max| slot per quantisation site (keyed by the consuming weight), two device arrays
    (`prev` = what this step quantises against, `cur` = what this step observes).  `advance()` rolls them over at the start of a step -- a
    kernel, so it is part of the captured graph.

    The producer-side shortcuts (an e4m3 twin written by the kernel that produces an activation, a gradient |max| recorded by the kernel that produces the
    gradient into the site's `cur` slot) rely on advance() -- it zeroes `cur` and forgets unclaimed twins -- so they are taken only between advance()
    and end_step() (`step_active`: the step harness, engine.TrainStep).  Any other loop (the reference's eager train.py sequence, several backward passes
    per optimizer step) quantises with the self-contained launches lavt_fp8_quantize / lavt_fp8_quantize_current and pins nothing."""
    SLOTS = 1024

    def __init__(self):
        self.step_active = False
        self.prev = self.cur = None
        self.slots = {}
        self.twins = {}          # (data_ptr, numel) of a bf16 tensor -> (its e4m3 twin written by the producing kernel, pointer of the |max| it was scaled by)
        self.dy_amax = {}        # (data_ptr, numel) of a gradient tensor -> pointer of its |max|, recorded by the producing kernel

    def _ensure(self, device):
        if self.prev is None or self.prev.device != device:
            self.prev = torch.zeros(self.SLOTS, dtype=torch.float32, device=device)
            self.cur = torch.zeros(self.SLOTS, dtype=torch.float32, device=device)
            self.slots = {}

    def slot(self, key, device):
        self._ensure(device)
        i = self.slots.get(key)
        if i is None:
            i = self.slots[key] = len(self.slots)
            if i >= self.SLOTS:
                raise RuntimeError("fp8: more quantisation sites than amax slots")
        return i

    def advance(self):
        self.twins.clear()
        self.dy_amax.clear()
        self.step_active = True
        if self.prev is not None and self.slots:
            K.check(K.lib.lavt_fp8_advance(K.ptr(self.prev), K.ptr(self.cur), len(self.slots), K.stream()))

    def end_step(self):
        """the harness' step is over: unclaimed twins / gradient maxima are dropped (they would pin full-size activations), producers stop writing them"""
        self.twins.clear()
        self.dy_amax.clear()
        self.step_active = False

    def site_ptrs(self, key, device):
        """(pointer of the |max| this step quantises against, pointer of the |max| this step records) of a delayed-scaling site"""
        i = self.slot(key, device)
        return self.prev.data_ptr() + 4 * i, self.cur.data_ptr() + 4 * i

    # (an entry holds its tensor: the address cannot be handed to another tensor while the entry lives -- until it is picked up, or advance() at the next step)
    def put_twin(self, y, q, a_ptr):
        self.twins[(y.data_ptr(), y.numel())] = (q, a_ptr, y)

    def put_dy_amax(self, dx, a_ptr):
        self.dy_amax[(dx.data_ptr(), dx.numel())] = (a_ptr, dx)

    def quantize(self, x, key):
        """bf16 activation rows -> (uint8 e4m3 tensor of the same shape, pointer of the amax float it was scaled by)"""
        i = self.slot(key, x.device)
        x = x.contiguous()
        tw = self.twins.pop((x.data_ptr(), x.numel()), None)
        if tw is not None and tw[1] == self.prev.data_ptr() + 4 * i and tw[0].shape == x.shape:
            return tw[0], tw[1]          # the producing kernel wrote the twin against this site's scale (lavt_norm_apply_q8 / lavt_bilinear_fwd_q8): no launch
        q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        K.check(K.lib.lavt_fp8_quantize(K.dt(x.dtype), K.ptr(x), K.ptr(q), x.numel(), self.prev.data_ptr() + 4 * i, self.cur.data_ptr() + 4 * i, K.stream()))
        return q, self.prev.data_ptr() + 4 * i

    def quantize_current(self, x, key):
        """the same with CURRENT scaling (|max| of x itself, one extra read pass; lavt_fp8_quantize_current): for gradients, whose range moves from step to
        step and which an uncalibrated (scale 1) first step would flush to zero.  The |max| lives in the site's `prev` slot (advance() leaves it alone:
        the site never writes `cur`)."""
        i = self.slot(key, x.device)
        x = x.contiguous()
        q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        ent = self.dy_amax.pop((x.data_ptr(), x.numel()), None)
        a_ptr = ent[0] if ent is not None else None
        if a_ptr is not None:
            # the kernel that produced x recorded its |max| (lavt_norm_bwd_apply_amax, into the site's `cur` slot: advance() zeroes it): one pass instead of two
            K.check(K.lib.lavt_fp8_quantize(K.dt(x.dtype), K.ptr(x), K.ptr(q), x.numel(), a_ptr, None, K.stream()))
            return q, a_ptr
        K.check(K.lib.lavt_fp8_quantize_current(K.dt(x.dtype), K.ptr(x), K.ptr(q), x.numel(), self.prev.data_ptr() + 4 * i, K.stream()))
        return q, self.prev.data_ptr() + 4 * i


fp8 = _Proxy("fp8")
# Linear layers take the fp8 path from this many GEMM rows up; by default NONE does (the decoder's 3x3 convolutions, 54 % of the FLOPs, are the
# fp8 contractions).  Measured on MI355X, Swin-B 4x480x480 against the reference's fp32 run: convolutions only -- pixel agreement 0.974, mask IoU on
# decisive pixels 0.984, 18.73 ms/step (bf16 18.83); plus every Linear with >= 4096 rows (stages 0-1, qkv of stage 2) -- agreement 0.930, IoU on
# decisive pixels 0.753 and 19.31 ms/step: per-tensor e4m3 on the early backbone features costs accuracy and, at one quantisation launch per
# GEMM, time.  LAVT_FP8_LINEAR_MIN_ROWS=<rows> enables it for experiments.
_FP8_LINEAR_MIN_ROWS = int(os.environ.get("LAVT_FP8_LINEAR_MIN_ROWS", str(1 << 30)))
# e4m3 convolutions only where the problem fills the chip with 128x128 tiles (the pipelined fp8 kernel's smallest): below that -- the 30x30 maps of decoder
# level 4 -- the fp8 launch falls to gemm_v2's 64x64 K loop (55 us at 4x30x30 against 44 us for the bf16 pipelined kernel) and buys nothing
_FP8_CONV_MIN_TILES = int(os.environ.get("LAVT_FP8_CONV_MIN_TILES", "200"))


_FP8_WGRAD = os.environ.get("LAVT_FP8_WGRAD", "1") != "0"          # e4m3 weight gradients of the fp8 convolutions (0: the bf16 fused-tap kernel)


def _fp8_conv_fills(M, N):
    return -(-M // 128) * -(-N // 128) >= _FP8_CONV_MIN_TILES


_FP8_TWINS = os.environ.get("LAVT_FP8_TWINS", "1") != "0"          # producers write the e4m3 twin / record the gradient |max| (0: separate quantiser launches)


def fp8_act_site(weight, M, C1, C2=0):
    """the quantisation site of the activations a 3x3 convolution with `weight` contracts in e4m3, or None when that convolution stays in bf16: what a
    producer of its first input passes to `bilinear(..., fp8_site=)` / `batch_norm_relu(..., fp8_site=)` to write the e4m3 twin itself"""
    if not (_FP8_TWINS and fp8_enabled() and C1 % 16 == 0 and C2 % 16 == 0 and _fp8_conv_fills(M, weight.shape[0])):
        return None
    return id(weight)


def fp8_dy_site(weight, M, C1):
    """the site of the output gradient of the convolution with `weight` when its backward quantises it (e4m3 data gradient), else None"""
    if not (_FP8_TWINS and fp8_enabled() and _FP8_DGRAD and weight.shape[0] % 16 == 0 and _fp8_conv_fills(M, C1)):
        return None
    return (id(weight), "dy")


# data gradients of the decoder's convolutions in e4m3 as well (dY quantised with current scaling against its own |max|); 0 = bf16 data gradients
_FP8_DGRAD = os.environ.get("LAVT_FP8_DGRAD", "1") != "0"


class _GradSinks:
    """Fused gradient accumulation: when a parameter has a sink (an fp32 view of a flat gradient buffer that the step
    harness zeroes once per step), weight-gradient kernels accumulate straight into it and the autograd Function returns
    None for that parameter -- no per-parameter zero-fill, no AccumulateGrad add.  `on_ready(param)` tells the bucketed
    all-reduce that the parameter's gradient is complete.  Without sinks the ops return ordinary gradient tensors."""

    def __init__(self):
        self.map = {}
        self.on_ready = None
        self.used = set()
        self.by_ptr = {}
        self.assigned = set()

    def set(self, params, on_ready=None):
        self.map = {id(p): (weakref.ref(p), p.grad) for p in params if p.grad is not None}
        self.by_ptr = {ent[1].data_ptr(): pid for pid, ent in self.map.items()}
        self.on_ready = on_ready
        self.used = set()
        self.assigned = set()

    def clear(self):
        self.map, self.on_ready, self.used = {}, None, set()
        self.by_ptr, self.assigned = {}, set()

    def begin_step(self):
        """the flat gradient buffer has just been zeroed (GradBuckets.zero): every parameter may receive ONE weight gradient until the next call"""
        self.used = set()
        self.assigned = set()

    def mark_assigned_ptr(self, ptr):
        """the launch that has just been enqueued OVERWRITES the whole gradient buffer at `ptr` with plain stores (a grouped weight-gradient member, the
        fused-tap convolution weight gradient): such a parameter does not need the step's zero fill (GradBuckets.set_zero_skip, engine.TrainStep)"""
        pid = self.by_ptr.get(ptr) if ptr else None
        if pid is not None:
            self.assigned.add(pid)

    def buf(self, p, shape):
        """-> (fp32 buffer of `shape` to accumulate into, is_sink)"""
        ent = self.map.get(id(p)) if p is not None else None
        if ent is not None and ent[0]() is p:
            # Sink targets are written with plain stores when a reduction is not split (gemm_v2.hip: `atomic = nsplit > 1 || accumulate`) and
            # each parameter is reported ready once per step: a parameter used twice in one forward, or micro-batch accumulation without
            # GradBuckets.zero() in between, would silently lose a contribution.  Refuse instead.
            if id(p) in self.used:
                raise RuntimeError("fused gradient accumulation: a second weight gradient for the same parameter within one step (parameter shared "
                                   "between two ops, or gradient accumulation over micro-batches) is not supported by the step harness; "
                                   "call ops.sinks.clear() to use ordinary autograd accumulation")
            self.used.add(id(p))
            return ent[1].view(shape), True
        return torch.zeros(shape, dtype=torch.float32, device=p.device), False

    def done(self, p, buf, is_sink):
        """value to return from backward for parameter p"""
        if not is_sink:
            return buf.view(p.shape)
        if self.on_ready is not None:
            self.on_ready(p)
        return None


sinks = _Proxy("sinks")


def _note(shape, flops=0.0, nbytes=0.0):
    """profiler annotation of the next C-ABI launch (bench.py's in-process family timing): problem shape + algorithmic work -- flops for the
    contractions, bytes (one read / write of every operand) for the HBM-bound kernels"""
    if K.prof.enabled:
        K.prof.note = {"flops": flops, "bytes": nbytes, "shape": shape}


def _f32(p: Optional[torch.Tensor]):
    if p is None:
        return None
    assert p.dtype == torch.float32 and p.is_contiguous()
    return p.detach()


_ZERO_PAGES = {}


def _scratch(n_floats, device):
    """fp32 scratch for two-stage reductions (fresh per call: stream-ordered reuse is the allocator's job)."""
    return torch.empty(int(n_floats), dtype=torch.float32, device=device)


_PWAM_RECORDS = os.environ.get("LAVT_PWAM_RECORDS", "1") != "0"          # A/B switch: 0 = the round-5 launch sequence (word-side reductions as TN launches + their reduction launches)
_TN_PARTIALS_MINK = int(os.environ.get("LAVT_TN_PARTIALS_MINK", "2048"))


def _tn_parts(n_floats, device):
    """One persistent fp32 buffer per device for the partial tiles of split weight-gradient reductions: a launch's pieces are consumed by its own
    reduction kernel before the next launch of the stream writes them again (stream order), so consecutive launches share it."""
    t = _ctx.tn_parts.get(device)
    if t is None or t.numel() < n_floats:
        t = _ctx.tn_parts[device] = torch.empty(max(int(n_floats), 4 << 20), dtype=torch.float32, device=device)
    return t




def _discard_out(n_floats, device):
    """an output nobody reads (the C of a column-sum-only weight-gradient member); its own buffer: the shared scratch may hold partial sums"""
    b = _ctx.sink_out.get(device)
    if b is None or b.numel() < n_floats:
        b = _ctx.sink_out[device] = torch.empty(n_floats, dtype=torch.float32, device=device)
    return b


class _ZeroArena:
    """Small buffers that must START AS ZEROS (targets of split reductions in the fused PWAM node, padded key / value rows) were one torch.zeros each:
    ~12 fill kernels of ~5 us per step, each a node of the captured chain.  Under the step harness they are carved from ONE arena whose used prefix
    (the high-water mark of the previous step) is zeroed by a single fill at the start of the step; a request the prefix cannot serve -- the first
    step, a different call sequence, a call outside the harness -- falls back to torch.zeros.  A buffer lives until the end of its step only."""
    BYTES = int(os.environ.get("LAVT_ZERO_ARENA_MB", "64")) << 20

    def __init__(self):
        self.buf, self.off, self.hw, self.zeroed, self.active = None, 0, 0, 0, False

    def begin_step(self, device, defer=False):
        """defer: the used prefix is returned (as a float32 view) for the caller to zero with its own fill -- the step harness hands it to the gradient
        buffer's multi-tensor zero, one launch for both"""
        if os.environ.get("LAVT_ZERO_ARENA", "1") == "0":
            return None
        if self.buf is None or self.buf.device != device:
            self.buf = torch.empty(self.BYTES, dtype=torch.uint8, device=device)
            self.hw = 0
        self.zeroed = min(self.hw, self.BYTES)
        view = None
        if self.zeroed:
            if defer:
                view = self.buf[:self.zeroed].view(torch.float32)          # (requests are rounded to 256 bytes)
            else:
                self.buf[:self.zeroed].zero_()
        self.off, self.hw, self.active = 0, 0, True
        return view

    def end_step(self):
        self.active = False

    def take(self, shape, dtype, device):
        n = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            n *= int(d)
        nbytes = -(-n * torch.empty(0, dtype=dtype).element_size() // 256) * 256
        if self.active and self.buf is not None and self.buf.device == device:
            start = self.off
            self.off += nbytes
            self.hw = max(self.hw, self.off)
            if start + nbytes <= self.zeroed:
                return self.buf[start:start + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(shape)
        return torch.zeros(shape, dtype=dtype, device=device)


zero_arena = _Proxy("zero_arena")


class _FillRiders:
    """The zero fill of the step's flat gradient buffer (475 MB for Swin-B: 58 us at the HBM rate) is needed by nothing before backward, yet as one
    launch it headed the captured chain.  Under the step harness it is handed over here (begin) and consumed in 32 MB slices by the fused W-MSA forward
    launches, which zero their slice with rider workgroups (lavt_wmsa_fwd_rider); finish() -- called between forward and backward -- zeroes whatever is
    left with an ordinary fill (all of it for a model without such launches)."""
    CHUNK = int(os.environ.get("LAVT_FILL_CHUNK_MB", "32")) << 20

    def __init__(self):
        self.buf, self.off = None, 0
        # measured (round 4): 8.311 / 8.321 ms per step with the riders against 8.328 without -- a 32 MB slice of writes per launch slows the DMA-latency-bound
        # forward launch by about what the stand-alone fill costs (the binning and LayerNorm riders read 7-13 MB per launch and are free).  Off by default.
        self.enabled = os.environ.get("LAVT_FILL_RIDERS", "0") == "1"

    def begin(self, flat):
        """-> True when the buffer's zero fill has been taken over (the caller must not fill it itself)"""
        if not self.enabled or not flat.is_cuda:
            return False
        self.buf, self.off = flat.view(torch.uint8), 0
        return True

    def take(self):
        """-> (address, bytes) of the next slice to zero inside a launch, or (None, 0)"""
        if self.buf is None:
            return None, 0
        n = min(self.CHUNK, ((self.buf.numel() - self.off) // 16) * 16)
        if n <= 0:
            return None, 0
        ptr = self.buf.data_ptr() + self.off
        self.off += n
        return ptr, n

    def finish(self):
        if self.buf is not None:
            if self.off < self.buf.numel():
                self.buf[self.off:].zero_()
            self.buf, self.off = None, 0


fill_riders = _Proxy("fill_riders")


def _zero_page_tensor(device):
    _zero_page(device)
    return _ZERO_PAGES[device]


def _zero_page(device):
    """256 zero bytes per device: source of every chunk that must read 0 in the LDS-DMA GEMM (padding rows, conv halo, tails)."""
    z = _ZERO_PAGES.get(device)
    if z is None:
        z = _ZERO_PAGES[device] = torch.zeros(64, dtype=torch.float32, device=device)
    return z.data_ptr()


# ------------------------------------------------------------------------------------------ raw launches
def gemm_nt(dtype, M, N, Kd, A, lda, B, ldb, Cout, ldc, *, batch=1, strideA=0, strideB=0, strideC=0, A2=None, lda2=0,
            a_split=0, a_rowmap=None, conv=None, b_kmajor=False, b_tap_stride=0, alpha=1.0, bias=None, strideBias=0,
            row_scale=None, strideRowScale=0, row_scale_div=1, act=K.ACT_NONE, Cpre=None, ldcpre=0, R=None, ldr=0, C2=None, ldc2=0,
            c_split=0, c_rowmap=None, c_f32=False, a_off=0, b_off=0, c_off=0, dact_pre=None, lddact=0, dact=K.ACT_NONE, deq=None,
            mul=None, ldmul=0, res_first=False, conv_tap_split=0, conv_kc_split=0, ln=None, want_colstats=False):
    """A/B/Cout are tensors; *_off are element offsets into them (column sub-blocks).  dtype torch.uint8 = e4m3 operands (A, B 1 byte per
    element; C / residual bf16) with the two dequantisation |max| pointers in `deq`.
    want_colstats: if this problem's kernel has the statistics epilogue (lavt_gemm_nt_colstats_plan), the launch also stores per-row-block column
    sums / centred second moments of the output; returns (partials, blocks, rows_per_block) then, else None."""
    f8 = dtype == torch.uint8
    es = 4 if dtype == torch.float32 else (1 if f8 else 2)
    p = K.GemmNT()
    p.dtype, p.M, p.N, p.K, p.batch = (K.FP8 if f8 else K.dt(dtype)), M, N, Kd, batch
    p.A, p.lda, p.strideA = K.ptr(A) + a_off * es, lda, strideA
    p.A2, p.lda2, p.a_split = K.ptr(A2), lda2, a_split
    p.a_rowmap = K.ptr(a_rowmap)
    if conv is not None:
        p.conv_h, p.conv_w, p.conv_kc, p.conv_flip = conv[:4]
        if len(conv) > 4:
            p.conv_d, p.conv_kd, p.conv_kh, p.conv_kw = conv[4:8]
    p.B, p.ldb, p.strideB, p.b_kmajor, p.b_tap_stride = K.ptr(B) + b_off * es, ldb, strideB, int(b_kmajor), b_tap_stride
    p.alpha, p.bias, p.strideBias = alpha, K.ptr(bias), strideBias
    p.row_scale, p.strideRowScale, p.row_scale_div, p.act = K.ptr(row_scale), strideRowScale, row_scale_div, act
    p.Cpre, p.ldcpre, p.R, p.ldr = K.ptr(Cpre), ldcpre, K.ptr(R), ldr
    p.C, p.ldc, p.strideC = K.ptr(Cout) + c_off * (4 if c_f32 else (2 if f8 else es)), ldc, strideC
    p.C2, p.ldc2, p.c_split = K.ptr(C2), ldc2, c_split
    p.c_rowmap, p.c_f32 = K.ptr(c_rowmap), int(c_f32)
    p.zeros = _zero_page(A.device)
    p.dact_pre, p.lddact, p.dact = K.ptr(dact_pre), lddact, dact
    p.mul, p.ldmul, p.res_first, p.conv_tap_split, p.conv_kc_split = K.ptr(mul), ldmul, int(res_first), conv_tap_split, conv_kc_split
    if ln is not None:          # (wsum, mean out, rstd out, eps): LayerNorm-folded A operand
        p.ln_wsum, p.ln_mean, p.ln_rstd, p.ln_eps = K.ptr(ln[0]), K.ptr(ln[1]), K.ptr(ln[2]), ln[3]
    if deq is not None:
        p.deq_a, p.deq_b = deq
    stats = None
    if want_colstats:
        rpb = C.c_int32(0)
        nblk = int(K.lib.lavt_gemm_nt_colstats_plan(C.byref(p), C.byref(rpb)))
        if nblk > 0:
            parts = torch.empty(nblk * 2 * N, dtype=torch.float32, device=A.device)
            p.colstats = K.ptr(parts)
            stats = (parts, nblk, int(rpb.value))
    if K.prof.enabled:
        K.prof.note = {"flops": 2.0 * M * N * Kd * batch, "shape": f"nt {M}x{N}x{Kd}" + (f" b{batch}" if batch > 1 else "") + (" conv" if conv is not None else "")
                       + (" kmajor" if b_kmajor else "")}
    K.check(K.lib.lavt_gemm_nt(C.byref(p), K.stream()))
    return stats


def gemm_tn(dtype, I, J, Kd, A, lda, B, ldb, Cout, ldc, *, batch=1, strideA=0, strideB=0, strideC=0, a_rowmap=None,
            a_rowscale=None, a_rowscale_div=1, a_rowscale_binary=False, accumulate=False, B2=None, ldb2=0, b_split=0, b_rowmap=None, conv=None, alpha=1.0, c_conv_permute=False, colsum=None,
            strideColsum=0, a_off=0, b_off=0, c_off=0, defer=None, colsum_atomic=False, extra=False, rider=None):
    es = 4 if dtype == torch.float32 else 2
    assert Cout.dtype == torch.float32
    p = K.GemmTN()
    p.dtype, p.I, p.J, p.K, p.batch = K.dt(dtype), I, J, Kd, batch
    p.A, p.lda, p.strideA, p.a_rowmap = K.ptr(A) + a_off * es, lda, strideA, K.ptr(a_rowmap)
    p.a_rowscale, p.a_rowscale_div = K.ptr(a_rowscale), a_rowscale_div
    p.a_rowscale_binary, p.accumulate = int(a_rowscale_binary), int(accumulate)
    p.B, p.ldb, p.strideB = K.ptr(B) + b_off * es, ldb, strideB
    p.B2, p.ldb2, p.b_split, p.b_rowmap = K.ptr(B2), ldb2, b_split, K.ptr(b_rowmap)
    if conv is not None:
        p.conv_h, p.conv_w, p.conv_kc = conv[:3]
        if len(conv) > 3:
            p.conv_d, p.conv_kd, p.conv_kh, p.conv_kw = conv[3:7]
    p.alpha, p.C, p.ldc, p.strideC = alpha, K.ptr(Cout) + c_off * 4, ldc, strideC
    p.c_conv_permute, p.split_k = int(c_conv_permute), (-1 if defer is not None else 0)     # deferred = into the zeroed flat gradient buffer: a grouped launch may split K
    p.colsum, p.strideColsum, p.colsum_atomic = K.ptr(colsum), strideColsum, int(colsum_atomic)
    # rows a row map may name (the pipelined grouped kernel bounds its 32-bit descriptor offsets with them)
    p.a_src_rows = (A.shape[0] if A.dim() >= 2 else A.numel() // max(lda, 1)) if a_rowmap is not None else 0
    p.b_src_rows = (B.shape[0] if B.dim() >= 2 else B.numel() // max(ldb, 1)) if b_rowmap is not None else 0
    p.zeros = _zero_page(A.device)
    if dtype == torch.bfloat16 and conv is None and (batch == 1 or defer is None) and Kd >= (_TN_PARTIALS_MINK if batch == 1 else 128) and os.environ.get("LAVT_TN_PARTIALS", "1") != "0":      # (one scratch per device: every launch of a step is on ONE stream)
        # scratch for split reductions through partial tiles (long-K weight gradients on few output tiles: PWAM's 1x1 convolutions over 28 800 rows;
        # batched: the per-sample word-side matrices of the fused PWAM node -- plain stores + a fixed-order sum, so the result is run-to-run identical)
        need = batch * int(K.lib.lavt_gemm_tn_pieces(C.byref(p))) * (I * J + I)
        if need <= (16 << 20):
            scr = _tn_parts(need, A.device)
            p.partials, p.partials_floats = K.ptr(scr), (need if defer is not None else scr.numel())     # queued: flush() hands every member its own region
    if defer is not None:
        if rider is not None:
            defer.add(p, (A, B, Cout, a_rowmap, a_rowscale, b_rowmap, colsum) + tuple(t for t in rider if torch.is_tensor(t)), extra=extra, rider=rider)
        else:
            defer.add(p, (A, B, Cout, a_rowmap, a_rowscale, b_rowmap, colsum), extra=extra)
        return
    if K.prof.enabled:
        K.prof.note = {"flops": 2.0 * I * J * Kd * batch, "shape": f"tn {I}x{J}x{Kd}" + (f" b{batch}" if batch > 1 else "") + (" conv" if conv is not None else "")}
    K.check(K.lib.lavt_gemm_tn(C.byref(p), K.stream()))


def assign_partials(items, device):
    """the members of one grouped launch write their partial tiles at the same time: disjoint regions of the per-device scratch"""
    need = [int(q.partials_floats) if q.partials else 0 for q in items]
    if sum(need):
        base, off = _tn_parts(sum(need), device).data_ptr(), 0
        for q, nf in zip(items, need):
            if nf:
                q.partials = base + 4 * off
                off += nf


_LN_RIDER = os.environ.get("LAVT_LN_RIDER", "1") != "0"


def _launch_ln_partial(rider):
    dy_, x_, g_, mean_, rstd_, dx_, ws_, dres_, rows_, C_ = rider
    K.check(K.lib.lavt_layernorm_bwd_partial(K.dt(x_.dtype), K.ptr(dy_), K.ptr(x_), None, K.ptr(g_), K.ptr(mean_), K.ptr(rstd_), K.ptr(dx_), K.ptr(ws_), ws_.numel(),
                                             K.ptr(dres_), rows_, C_, K.stream()))


# measured (round 4, tools/wgrad_sk_time.py): 70-78 us against 39.6 us for the 64x64-tile launch of the stage-2 block -- contiguous runs lose the L2 / MALL sharing
# of operand panels between workgroups that sweep K together.  Off by default; the launch stays reachable for experiments.
_STREAMK = os.environ.get("LAVT_WGRAD_STREAMK", "0") == "1"


class _WgradQueue:
    """Weight-gradient GEMMs of consecutive backward ops (the four Linear layers of a Swin block) are collected and issued as ONE grouped
    launch (lavt_gemm_tn_grouped): together they fill the chip without split-K, so the gradients are stored plainly instead of going
    through fp32 atomics.  Only used when gradients accumulate into the step harness' flat buffer (ops.sinks) in `exclusive` mode: every
    parameter receives exactly one weight gradient per step into a buffer that was zeroed at the start of the step, so store == accumulate.
    `on_ready` notifications (DDP buckets) are held back until the group has been enqueued."""

    def __init__(self):
        self.enabled = False
        self.items, self.keep, self.ready, self.scopes, self.primary = [], [], [], [], 0
        self.pending = set()         # ids of parameters whose weight gradient is still queued (their autograd hook fires before the launch)

    def active(self):
        return self.enabled and sinks.map and os.environ.get("LAVT_WGRAD_GROUP", "1") != "0"

    def add(self, p, tensors, extra=False, rider=None):
        """extra: a side member (at most two per group: lavt_gemm_tn_grouped takes six problems) that does not count towards the four-member flush.
        rider: (dy, x, gamma, mean, rstd, dx, ws, dres, rows, C) of a LayerNorm backward (partial-sum form) that must be launched with or right after
        this member: if the member completes the group, the LayerNorm rides as extra workgroups of the grouped launch (lavt_gemm_tn_grouped_ln);
        otherwise it is launched on its own at once."""
        self.items.append(p)
        self.keep.append(tensors)
        self.scopes.append(K.prof.label)
        self.primary += 0 if extra else 1
        # a member that brings a rider closes the group: it is the last weight gradient of its Swin block (qkv), so groups are the block's own four
        # members + side member -- without this the four-member count ran one member out of phase (fc1, proj, side, qkv of a block + fc2 of the next)
        # and the launch was issued from the next block's MLP backward, where no LayerNorm waits to ride
        if self.primary == 4 or len(self.items) == 6 or (rider is not None and _LN_RIDER):
            self.flush(rider)
        elif rider is not None:
            _launch_ln_partial(rider)

    def notify(self, param):
        # The op returns None for this parameter's gradient, and PyTorch runs the parameter's post-accumulate hook all the same -- BEFORE the
        # grouped kernel exists.  Until flush() has enqueued it the parameter is `pending`: GradBuckets ignores hook reports for pending
        # parameters, or a bucket whose last member is a queued gradient would be all-reduced while that gradient is still being written.
        self.ready.append(param)
        self.pending.add(id(param))

    def flush(self, rider=None):
        if not self.items and rider is not None:
            _launch_ln_partial(rider)
        if self.items:
            if not _STREAMK:             # grouped members store plainly (direct tiles and the reduction of partial tiles alike) unless asked to accumulate
                for q in self.items:
                    if not q.accumulate:
                        sinks.mark_assigned_ptr(q.C)
            assign_partials(self.items, next(t for t in self.keep[0] if t is not None).device)
            arr = (K.GemmTN * len(self.items))(*self.items)
            if K.prof.enabled:       # a grouped launch mixes scopes (qkv / proj with fc1 / fc2): the note carries the per-member flops and labels
                fl = [2.0 * q.I * q.J * q.K for q in self.items]
                K.prof.note = {"flops": sum(fl), "shape": "tn-grouped " + "+".join(f"{q.I}x{q.J}x{q.K}" for q in self.items),
                               "members": list(zip(self.scopes, fl))}
            n_items = len(self.items)
            tensors = [t for tup in self.keep for t in tup if t is not None]
            sk = int(K.lib.lavt_gemm_tn_grouped_sk_ws(arr, n_items)) if _STREAMK else 0
            if sk:
                # stream-K form (csrc/gemm_tn_v2.hip): 128x128 tiles, equal runs of K-tile iterations per persistent workgroup, split tiles through a scratch
                dev = tensors[0].device
                scr = _ctx.sk_scratch.get(dev)
                if scr is None or scr.numel() < sk:
                    scr = _ctx.sk_scratch[dev] = torch.empty(sk, dtype=torch.float32, device=dev)
                K.check(K.lib.lavt_gemm_tn_grouped_sk(arr, n_items, K.ptr(scr), scr.numel(), K.stream()))
                if rider is not None:
                    _launch_ln_partial(rider)
                self._after_flush()
                return
            if rider is not None and _LN_RIDER:
                dy_, x_, g_, mean_, rstd_, dx_, ws_, dres_, rows_, C_ = rider
                K.check(K.lib.lavt_gemm_tn_grouped_ln(arr, n_items, K.ptr(dy_), K.ptr(x_), K.ptr(g_), K.ptr(mean_), K.ptr(rstd_), K.ptr(dx_), K.ptr(ws_), ws_.numel(),
                                                      K.ptr(dres_), rows_, C_, K.stream()))
                self._after_flush()
                return
            K.check(K.lib.lavt_gemm_tn_grouped(arr, n_items, K.stream()))
            if rider is not None:
                _launch_ln_partial(rider)
        self._after_flush()

    def _after_flush(self):
        ready = self.ready
        self.items, self.keep, self.ready, self.scopes, self.primary = [], [], [], [], 0
        for prm in ready:
            self.pending.discard(id(prm))
            if sinks.on_ready is not None:
                sinks.on_ready(prm)


wgrads = _Proxy("wgrads")


_LN_REDUCE_COMPACT = os.environ.get("LAVT_LN_REDUCE_COMPACT", "1") != "0"          # A/B switch: 0 = the (128 column blocks x sets) grid of round 3


class _LnDeferred:
    """LayerNorm weight / bias gradients feed nothing but the gradient buffer, so under the step harness their per-workgroup partial sums are
    parked in a persistent arena and reduced by ONE launch at the end of backward (lavt_reduce_partials_multi) instead of one two-kernel
    reduction per LayerNorm (56 per Swin-B step, ~4.9 us each).  The arena offsets and the sink addresses repeat from step to step, so the
    device descriptor table is built once (outside a capture) and reused; a step whose sequence differs rebuilds it (not possible while
    capturing: that raises)."""
    ARENA_FLOATS = int(os.environ.get("LAVT_LN_ARENA_M", "96")) << 20          # 384 MiB of the 288 GB: Swin-B needs 37 x 450 x 1024 + ... = 24 M floats, Video-Swin-B (4608 rows) 50 M; beyond it a LayerNorm reduces at once

    def __init__(self):
        self.arena, self.off, self.items, self.params, self.tables = None, 0, [], [], []
        self.desc, self.desc_key, self.tdesc, self.tdesc_key = None, None, None, None

    def active(self):
        return wgrads.active() and os.environ.get("LAVT_LN_DEFER", "1") != "0"

    def alloc(self, nfloats, device):
        if self.arena is None or self.arena.device != device:
            self.arena, self.off = torch.empty(self.ARENA_FLOATS, dtype=torch.float32, device=device), 0
        if self.off + nfloats > self.arena.numel():
            return None
        v = self.arena[self.off:self.off + nfloats]
        self.off += -(-nfloats // 64) * 64
        return v

    def add(self, ws, nblk, C, dg, db, params):
        self.items.append((ws.data_ptr(), nblk, C, dg.data_ptr(), db.data_ptr()))
        for p in params:
            self.params.append(p)
            wgrads.pending.add(id(p))          # its autograd hook fires now, before the reduction exists: GradBuckets ignores pending parameters

    def add_table(self, parts, pieces, heads, R, dtable, param):
        """attention bias-table gradient: per-workgroup histograms parked in the arena (lavt_attn_dtable_finish_multi at flush)"""
        self.tables.append((parts.data_ptr(), pieces, heads, R, dtable.data_ptr()))
        self.params.append(param)
        wgrads.pending.add(id(param))

    def flush(self):
        dtable_chain.flush()                     # the last attention-backward launch's binning job (chained form) runs on its own
        if self.tables:
            key = tuple(self.tables)
            if key != self.tdesc_key:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("deferred table gradients: the step being captured differs from the warm-up steps")
                self.tdesc = torch.tensor(self.tables, dtype=torch.int64).to(self.arena.device)
                self.tdesc_key = key
            if len(self.tables) <= 64 and _LN_REDUCE_COMPACT:
                K.check(K.lib.lavt_attn_dtable_finish_multi_compact(K.ptr(self.tdesc), len(self.tables), max(t[3] for t in self.tables), sum(t[2] for t in self.tables), K.stream()))
            else:
                K.check(K.lib.lavt_attn_dtable_finish_multi(K.ptr(self.tdesc), len(self.tables), max(t[3] for t in self.tables), max(t[2] for t in self.tables), K.stream()))
            self.tables = []
        if self.items:
            key = tuple(self.items)
            if key != self.desc_key:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("deferred LayerNorm reductions: the step being captured differs from the warm-up steps (descriptor table would need a host copy)")
                self.desc = torch.tensor(self.items, dtype=torch.int64).to(self.arena.device)
                self.desc_key = key
            K.check(K.lib.lavt_reduce_partials_multi(K.ptr(self.desc), len(self.items), sum(int(K.lib.lavt_reduce_partials_column_blocks(it[2])) for it in self.items) if _LN_REDUCE_COMPACT else 0, K.stream()))
        params = self.params
        self.items, self.params, self.off = [], [], 0
        for p in params:
            wgrads.pending.discard(id(p))
            if sinks.on_ready is not None:
                sinks.on_ready(p)


ln_deferred = _Proxy("ln_deferred")


class _DtableChain:
    """Chained table-gradient binning (lavt_window_attn_bwd_chained): an attention-backward launch leaves the binning of its dS slabs to a LATER launch
    of the backward pass, which runs it as extra workgroups -- the next LayerNorm backward of an MLP (round 6: 256-thread workgroups without LDS
    pressure; lavt_layernorm_bwd_partial_xn_dtable), else the next attention-backward launch (round 4); the last job of a pass is launched on its
    own from ln_deferred.flush().  The slabs (and the histogram scratch) of a pending job are kept alive here until it has run."""

    def __init__(self):
        self.job, self.keep = None, None
        self.enabled = os.environ.get("LAVT_DTABLE_CHAIN", "1") != "0"
        self.ln_host = os.environ.get("LAVT_DTABLE_LN_HOST", "1") != "0"          # A/B switch: 0 = riders only in attention-backward launches

    def take_for_layernorm(self):
        """the pending job for a LayerNorm-backward launch to carry (None: nothing pending / switched off); the caller reports back with done_by_layernorm()"""
        return self.job if (self.job is not None and self.ln_host) else None

    def done_by_layernorm(self):
        self.job, self.keep = None, None

    def launch(self, dtype, qkv, ld, region, nw_img, out, dout, lse, dqkv, table, wsb, parts, wd, wh, ww, nwin, N, heads, hd, scale):
        mine = K.DtableJob()
        prev = C.byref(self.job) if self.job is not None else None
        K.check(K.lib.lavt_window_attn_bwd_chained(K.dt(dtype), K.ptr(qkv), ld, K.ptr(region), nw_img, K.ptr(out), K.ptr(dout), K.ptr(lse), K.ptr(dqkv), K.ptr(_f32(table)),
                                                   K.ptr(wsb), wsb.numel(), K.ptr(parts), wd, wh, ww, nwin, N, heads, hd, scale, prev, C.byref(mine), K.stream()))
        self.job, self.keep = mine, (wsb, parts)

    def flush(self):
        if self.job is not None:
            K.check(K.lib.lavt_attn_dtable_run(C.byref(self.job), K.stream()))
            self.job, self.keep = None, None


dtable_chain = _Proxy("dtable_chain")


def lang_mask(l_mask, B, n_l):
    """(B, n_l[, 1]) 0 / 1 language mask -> (mask_rows [B * n_l] float32, maskbias [B, KV_LD] = 1e4 * m - 1e4 with -1e4 in the padding slots)"""
    m = l_mask.reshape(B, n_l)
    if m.dtype not in (torch.float32, torch.int64):
        m = m.to(torch.float32)
    m = m.contiguous()
    rows = torch.empty(B * n_l, dtype=torch.float32, device=m.device)
    bias = torch.empty(B, KV_LD, dtype=torch.float32, device=m.device)
    K.check(K.lib.lavt_lang_mask(K.ptr(m), int(m.dtype == torch.int64), K.ptr(rows), K.ptr(bias), B, n_l, KV_LD, K.stream()))
    return rows, bias


def droppath_factors(u, keep):
    """u [n, B] uniform draws, keep [n, 1] keep probabilities -> floor(keep + u) / keep (timm's drop_path factors), one launch"""
    f = torch.empty_like(u)
    K.check(K.lib.lavt_droppath_factors(K.ptr(u), K.ptr(keep.contiguous()), K.ptr(f), u.shape[0], u.shape[1], K.stream()))
    return f




def droppath_draw(keep, B):
    """floor(keep + u) / keep [n, B] with u drawn on the device (Philox4x32-10; csrc/elementwise.hip): the generator state -- (seed, draw counter), seeded
    from torch's default generator when first used on a device -- lives in device memory and is advanced by the kernel, so a captured step draws fresh
    factors on every replay without torch.rand's two bookkeeping fills.  LAVT_DROPPATH_RNG=torch keeps torch.rand."""
    dev = keep.device
    st = _ctx.dp_state.get(dev)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7fffffffffffffff, 0], dtype=torch.int64).to(dev)
        _ctx.dp_state[dev] = st
    f = torch.empty(keep.shape[0], B, dtype=torch.float32, device=dev)
    K.check(K.lib.lavt_droppath_draw(K.ptr(st), K.ptr(keep.contiguous()), K.ptr(f), keep.shape[0], B, K.stream()))
    return f


def droppath_reseed(seed, device=None):
    """restart the device DropPath generator (all devices, or one) from `seed`.  The generator is seeded from torch.initial_seed() when FIRST used on a
    device: a later torch.manual_seed() does not reach it -- call this after seeding (every DDP rank with the same seed draws the same masks, as
    torch.rand under a common seed does)."""
    for dev, st in _ctx.dp_state.items():
        if device is None or torch.device(device) == dev:
            st.copy_(torch.tensor([int(seed) & 0x7fffffffffffffff, 0], dtype=torch.int64))


def droppath_state_dict():
    """{device string: (seed, draw counter)} of the device DropPath generators: put it into a checkpoint next to the optimizer state (the reference's
    torch.rand masks resume from the torch generator state; this generator lives outside it) and hand it to droppath_load_state_dict on resume"""
    return {str(dev): tuple(int(v) for v in st.cpu()) for dev, st in _ctx.dp_state.items()}


def droppath_load_state_dict(state):
    for dev, (seed, counter) in state.items():
        d = torch.device(dev)
        t = torch.tensor([int(seed), int(counter)], dtype=torch.int64)
        if d in _ctx.dp_state:
            _ctx.dp_state[d].copy_(t)
        else:
            _ctx.dp_state[d] = t.to(d)


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    K.check(K.lib.lavt_cast(K.dt(x.dtype), K.ptr(x.contiguous()), K.dt(dtype), K.ptr(out), x.numel(), K.stream()))
    return out


# ------------------------------------------------------------------------------------------ Linear (+gather/scatter/act/residual)
@dataclass
class LinOpts:
    act: int = K.ACT_NONE
    in_map: Optional[torch.Tensor] = None      # GEMM row m reads x[in_map[m]] (-1 = zeros)
    out_map: Optional[torch.Tensor] = None     # GEMM row m writes y[out_map[m]] (-1 = dropped)
    rows: Optional[int] = None                 # number of GEMM rows (default: x rows)
    out_rows: Optional[int] = None             # rows of y (default: GEMM rows)
    zero_init: bool = False                    # y rows not covered by out_map must read 0
    row_scale: Optional[torch.Tensor] = None   # fp32 factor of GEMM row m: row_scale[m // row_scale_div] (language mask, DropPath)
    row_scale_div: int = 1
    row_scale_value: float = 0.0               # if != 0: row_scale holds only 0 and this value (lets the wgrad kernel treat it as a row mask)
    out_inv: Optional[torch.Tensor] = None     # inverse of out_map on the output rows (out_inv[row of y] = GEMM row): the weight gradient then contracts
                                               # over the out_rows real rows instead of the GEMM rows (padded window positions drop out)


@K.scoped
class _Linear(torch.autograd.Function):
    """y[out_map[m]] = act((x[in_map[m]] @ W^T + b) * row_scale[m // div]) + residual[out_map[m]]"""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, o: LinOpts):
        x = x.contiguous()
        dtype = x.dtype
        Wc = weights.get(weight, dtype, "lin")
        N, Kd = Wc.shape
        assert x.shape[1] == Kd, f"linear: x has {x.shape[1]} channels, weight expects {Kd}"
        M = o.rows if o.rows is not None else x.shape[0]
        assert o.row_scale is None or o.act == K.ACT_NONE
        out_rows = o.out_rows if o.out_rows is not None else M
        y = (torch.zeros if o.zero_init else torch.empty)(out_rows, N, dtype=dtype, device=x.device)
        pre = torch.empty_like(y) if o.act != K.ACT_NONE else None
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == y.shape and residual.dtype == dtype
        if dtype == torch.bfloat16 and fp8_enabled() and Kd % 16 == 0 and M >= _FP8_LINEAR_MIN_ROWS:
            # configs[4]: e4m3 activations x e4m3 weights on the fp8 MFMA for the forward contraction (gathered rows stay row gathers; the zero
            # page is +0 in e4m3 as well); backward runs on the saved bf16 tensors
            Wq, w_amax = weights.get_fp8(weight, "lin")
            xq, a_ptr = fp8.quantize(x, id(weight))
            gemm_nt(torch.uint8, M, N, Kd, xq, Kd, Wq, Kd, y, N, a_rowmap=o.in_map, bias=_f32(bias), row_scale=o.row_scale, row_scale_div=o.row_scale_div,
                    act=o.act, Cpre=pre, ldcpre=N, R=residual, ldr=N, c_rowmap=o.out_map, deq=(a_ptr, w_amax.data_ptr()))
        elif (o.out_inv is not None and o.out_map is not None and o.in_map is None and out_rows < M and not o.zero_init
              and os.environ.get("LAVT_TOKEN_ORDER_DGRAD", "1") != "0"):
            # windowed rows scattered back to tokens (proj): one GEMM row per TOKEN, its input row gathered through the inverse map -- the rows of
            # padded window positions (dropped by the scatter) are never computed, and the output is written densely
            gemm_nt(dtype, out_rows, N, Kd, x, Kd, Wc, Kd, y, N, a_rowmap=o.out_inv, bias=_f32(bias), row_scale=o.row_scale,
                    row_scale_div=max(o.row_scale_div * out_rows // M, 1), act=o.act, Cpre=pre, ldcpre=N, R=residual, ldr=N)
        else:
            gemm_nt(dtype, M, N, Kd, x, Kd, Wc, Kd, y, N, a_rowmap=o.in_map, bias=_f32(bias), row_scale=o.row_scale, row_scale_div=o.row_scale_div, act=o.act,
                    Cpre=pre, ldcpre=N, R=residual, ldr=N, c_rowmap=o.out_map)
        ctx.o, ctx.M, ctx.has_res = o, M, residual is not None
        ctx.save_for_backward(x, weight, pre, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, pre, bias = ctx.saved_tensors
        o, M = ctx.o, ctx.M
        dtype = x.dtype
        dy = dy.contiguous()
        g = dy
        if o.act != K.ACT_NONE:
            g = torch.empty_like(dy)
            K.check(K.lib.lavt_act_bwd(K.dt(dtype), o.act, K.ptr(dy), K.ptr(pre), K.ptr(g), dy.numel(), K.stream()))
        Wc = weights.get(weight, dtype, "lin")
        N, Kd = Wc.shape
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=dtype, device=x.device)     # in_map scatters cover every source row exactly once
            gemm_nt(dtype, M, Kd, N, g, N, Wc, Kd, dx, Kd, a_rowmap=o.out_map, b_kmajor=True, row_scale=o.row_scale,
                    row_scale_div=o.row_scale_div, c_rowmap=o.in_map)
        if ctx.needs_input_grad[1]:
            wbuf, wsink = sinks.buf(weight, (N, Kd))
            bbuf = bsink = None
            if bias is not None and ctx.needs_input_grad[2]:
                bbuf, bsink = sinks.buf(bias, (N,))
            binary = o.row_scale is not None and o.row_scale_value != 0.0
            grouped = wgrads.active() and wsink and (bbuf is None or bsink) and dtype == torch.bfloat16 and (o.row_scale is None or binary)
            token_order = (o.out_inv is not None and o.out_map is not None and o.in_map is None and g.shape[0] < M and dtype == torch.bfloat16
                           and os.environ.get("LAVT_TOKEN_ORDER_WGRAD", "1") != "0")
            if token_order and grouped:
                # windowed rows scattered back to tokens (proj): dW = sum over the TOKENS of dy[t]^T x[inv[t]] -- the padded window positions (dy = 0
                # there) drop out of the reduction: K = tokens instead of window rows (1800 instead of 2592 at stage 2, 450 instead of 1152 at stage 3)
                T_ = g.shape[0]
                gemm_tn(dtype, N, Kd, T_, g, N, x, Kd, wbuf, Kd, a_rowscale=o.row_scale, a_rowscale_div=max(o.row_scale_div * T_ // M, 1) if o.row_scale is not None else 1,
                        a_rowscale_binary=binary, alpha=o.row_scale_value if binary else 1.0, b_rowmap=o.out_inv, colsum=bbuf, defer=wgrads)
                wgrads.notify(weight)
                if bbuf is not None:
                    wgrads.notify(bias)
                dW = db = None
            elif grouped:                         # joins the block's grouped launch; the parameters report ready when it is enqueued
                gemm_tn(dtype, N, Kd, M, g, N, x, Kd, wbuf, Kd, a_rowmap=o.out_map, a_rowscale=o.row_scale, a_rowscale_div=o.row_scale_div,
                        a_rowscale_binary=binary, alpha=o.row_scale_value if binary else 1.0, b_rowmap=o.in_map, colsum=bbuf, defer=wgrads)
                wgrads.notify(weight)
                if bbuf is not None:
                    wgrads.notify(bias)
                dW = db = None
            else:
                gemm_tn(dtype, N, Kd, M, g, N, x, Kd, wbuf, Kd, a_rowmap=o.out_map, a_rowscale=o.row_scale, a_rowscale_div=o.row_scale_div, a_rowscale_binary=binary,
                        alpha=o.row_scale_value if binary else 1.0, b_rowmap=o.in_map, colsum=bbuf)
                dW = sinks.done(weight, wbuf, wsink)
                if bbuf is not None:
                    db = sinks.done(bias, bbuf, bsink)
        d_res = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dW, db, d_res, None


@K.scoped
class _LinearCat(torch.autograd.Function):
    """[x W1^T + b1 | x W2^T + b2 | ...]: several Linear layers of the same input as ONE GEMM over the stacked weight (weights.get_cat) and
    ONE data-gradient GEMM; the weight / bias gradients stay per parameter (column blocks of dy)."""

    @staticmethod
    def forward(ctx, x, *wb):
        x = x.contiguous()
        dtype = x.dtype
        ws, bs = wb[0::2], wb[1::2]
        Wc = weights.get_cat(ws, dtype)
        Nt, Kd = Wc.shape
        M = x.shape[0]
        y = torch.empty(M, Nt, dtype=dtype, device=x.device)
        bias = weights.get_bias_cat(bs) if all(b is not None for b in bs) else None
        gemm_nt(dtype, M, Nt, Kd, x, Kd, Wc, Kd, y, Nt, bias=bias)
        ctx.save_for_backward(x, *wb)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *wb = ctx.saved_tensors
        ws, bs = wb[0::2], wb[1::2]
        dtype = x.dtype
        dy = dy.contiguous()
        Wc = weights.get_cat(ws, dtype)
        Nt, Kd = Wc.shape
        M = x.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dtype, M, Kd, Nt, dy, Nt, Wc, Kd, dx, Kd, b_kmajor=True)
        grads = []
        off = 0
        for w, b in zip(ws, bs):
            n = w.shape[0]
            wbuf, wsink = sinks.buf(w, (n, Kd))
            bbuf, bsink = sinks.buf(b, (n,)) if b is not None else (None, True)
            if wgrads.active() and wsink and bsink and dtype == torch.bfloat16:
                gemm_tn(dtype, n, Kd, M, dy, Nt, x, Kd, wbuf, Kd, colsum=bbuf, a_off=off, defer=wgrads)
                wgrads.notify(w)
                if b is not None:
                    wgrads.notify(b)
                grads += [None, None]
            else:
                gemm_tn(dtype, n, Kd, M, dy, Nt, x, Kd, wbuf, Kd, colsum=bbuf, a_off=off)
                grads += [sinks.done(w, wbuf, wsink), sinks.done(b, bbuf, bsink) if b is not None else None]
            off += n
        return (dx, *grads)


def linear_cat(x, layers):
    """layers: nn.Linear modules sharing the input x -> [M, sum out_features]"""
    wb = []
    for m in layers:
        wb += [m.weight, m.bias]
    return _LinearCat.apply(x, *wb)


def linear(x, weight, bias=None, residual=None, **kw):
    return _Linear.apply(x, weight, bias, residual, LinOpts(**kw))


# The LayerNorm-folded MLP node keeps GELU'(pre) instead of pre for backward (LAVT_ACT_GELU_D / LAVT_ACT_STORED): the derivative shares the
# activation's exponential in the fc1 epilogue (+2 fma per element), and the fc2 data gradient's epilogue becomes one multiply instead of an erf +
# two exponentials per element (5 us of vector math on a 10 us GEMM at stage 2, tools/mlp_gemm_probe.py).  Only that launch is built with the
# branch: compiled into every NT kernel's epilogue it made the whole step 0.16 ms slower (tools/ab_lib.sh).  LAVT_GELU_D=0: the pre-activation form.
_GELU_FWD, _GELU_BWD = (K.ACT_GELU_D, K.ACT_STORED) if os.environ.get("LAVT_GELU_D", "1") != "0" else (K.ACT_GELU, K.ACT_GELU)


@K.scoped
class _Mlp(torch.autograd.Function):
    """y = fc2(GELU(fc1(x))) * row_scale + residual (Swin Mlp, reference lib/backbone.py:24-30, with the block's DropPath and residual folded in)
    as ONE autograd node, so that backward can hand the gradient through the activation inside a GEMM: fc2's data-gradient GEMM applies
    GELU'(pre) in its epilogue (lavt_gemm_nt dact_pre) and directly produces d/d pre -- no element-wise GELU-backward pass, no extra [M, 4C]
    round trip.  bf16 only (the exact-fp32 path keeps the two-op form)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, o: LinOpts):
        x = x.contiguous()
        dtype = x.dtype
        W1, W2 = weights.get(w1, dtype, "lin"), weights.get(w2, dtype, "lin")
        Hd, Cin = W1.shape
        Cout = W2.shape[0]
        M = x.shape[0]
        pre = torch.empty(M, Hd, dtype=dtype, device=x.device)
        h = torch.empty_like(pre)
        gemm_nt(dtype, M, Hd, Cin, x, Cin, W1, Cin, h, Hd, bias=_f32(b1), act=K.ACT_GELU, Cpre=pre, ldcpre=Hd)
        y = torch.empty(M, Cout, dtype=dtype, device=x.device)
        if residual is not None:
            residual = residual.contiguous()
        gemm_nt(dtype, M, Cout, Hd, h, Hd, W2, Hd, y, Cout, bias=_f32(b2), row_scale=o.row_scale, row_scale_div=o.row_scale_div, R=residual, ldr=Cout)
        ctx.save_for_backward(x, w1, b1, w2, b2, pre, h)
        ctx.o, ctx.has_res = o, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2, pre, h = ctx.saved_tensors
        o = ctx.o
        dtype = x.dtype
        dy = dy.contiguous()
        W1, W2 = weights.get(w1, dtype, "lin"), weights.get(w2, dtype, "lin")
        Hd, Cin = W1.shape
        Cout = W2.shape[0]
        M = x.shape[0]
        # d pre = ((dy * row_scale) W2) * GELU'(pre): the activation gradient rides in the data-gradient GEMM's epilogue
        dpre = torch.empty_like(pre)
        gemm_nt(dtype, M, Hd, Cout, dy, Cout, W2, Hd, dpre, Hd, b_kmajor=True, row_scale=o.row_scale, row_scale_div=o.row_scale_div,
                dact_pre=pre, lddact=Hd, dact=K.ACT_GELU)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dtype, M, Cin, Hd, dpre, Hd, W1, Cin, dx, Cin, b_kmajor=True)
        binary = o.row_scale is not None and o.row_scale_value != 0.0
        grads = []
        for (w, b, g, inp, n, kd, rs) in ((w2, b2, dy, h, Cout, Hd, True), (w1, b1, dpre, x, Hd, Cin, False)):
            wbuf, wsink = sinks.buf(w, (n, kd))
            bbuf, bsink = sinks.buf(b, (n,)) if b is not None else (None, True)
            kw = dict(a_rowscale=o.row_scale, a_rowscale_div=o.row_scale_div, a_rowscale_binary=binary, alpha=o.row_scale_value if binary else 1.0) if rs and o.row_scale is not None else {}
            if wgrads.active() and wsink and bsink and (not kw or binary):
                gemm_tn(dtype, n, kd, M, g, n, inp, kd, wbuf, kd, colsum=bbuf, defer=wgrads, **kw)
                wgrads.notify(w)
                if b is not None:
                    wgrads.notify(b)
                grads += [None, None]
            else:
                gemm_tn(dtype, n, kd, M, g, n, inp, kd, wbuf, kd, colsum=bbuf, **kw)
                grads += [sinks.done(w, wbuf, wsink), sinks.done(b, bbuf, bsink) if b is not None else None]
        dw2, db2, dw1, db1 = grads
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res and ctx.needs_input_grad[5] else None), None


@K.scoped
class _LnMlp(torch.autograd.Function):
    """x + DropPath(fc2(GELU(fc1(LN(x))))) -- norm2 + Mlp + residual of a Swin block (reference lib/backbone.py:243-245, 24-30) as ONE autograd node
    whose forward has no LayerNorm launch: fc1 contracts the RAW rows with the gamma-folded weight and applies the normalisation in its epilogue
    (lavt_gemm_nt.ln_wsum: row statistics from the A tiles the GEMM streams anyway).  Backward: fc2 data gradient with GELU' in the epilogue ->
    fc1 data gradient -> LayerNorm backward, which also writes the LayerNorm output the forward never materialised (the fc1 weight gradient's
    operand) and adds the residual branch's gradient -> the two weight gradients (grouped)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, eps, o: LinOpts):
        x = x.contiguous()
        dtype, dev = x.dtype, x.device
        M, Cin = x.shape
        Wg, wsum, biasp = weights.get_lnfold(w1, b1, gamma, beta)
        W2 = weights.get(w2, dtype, "lin")
        Hd, Cout = Wg.shape[0], W2.shape[0]
        pre = torch.empty(M, Hd, dtype=dtype, device=dev)
        h = torch.empty_like(pre)
        st = torch.empty(2, M, dtype=torch.float32, device=dev)
        gemm_nt(dtype, M, Hd, Cin, x, Cin, Wg, Cin, h, Hd, bias=biasp, act=_GELU_FWD, Cpre=pre, ldcpre=Hd, ln=(wsum, st[0], st[1], eps))
        y = torch.empty(M, Cout, dtype=dtype, device=dev)
        gemm_nt(dtype, M, Cout, Hd, h, Hd, W2, Hd, y, Cout, bias=_f32(b2), row_scale=o.row_scale, row_scale_div=o.row_scale_div, R=x, ldr=Cout)
        ctx.save_for_backward(x, gamma, beta, w1, b1, w2, b2, pre, h, st)
        ctx.o = o
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, w1, b1, w2, b2, pre, h, st = ctx.saved_tensors
        o = ctx.o
        dtype, dev = x.dtype, x.device
        dy = dy.contiguous()
        W1, W2 = weights.get(w1, dtype, "lin"), weights.get(w2, dtype, "lin")
        Hd, Cin = W1.shape
        Cout = W2.shape[0]
        M = x.shape[0]
        dpre = torch.empty_like(pre)
        gemm_nt(dtype, M, Hd, Cout, dy, Cout, W2, Hd, dpre, Hd, b_kmajor=True, row_scale=o.row_scale, row_scale_div=o.row_scale_div,
                dact_pre=pre, lddact=Hd, dact=_GELU_BWD)
        dxn = torch.empty_like(x)
        gemm_nt(dtype, M, Cin, Hd, dpre, Hd, W1, Cin, dxn, Cin, b_kmajor=True)
        # LayerNorm backward: dx = LN'(dxn) + dy (the residual branch), xn written on the way
        dx = torch.empty_like(x)
        xn = torch.empty_like(x)
        dg, gs = sinks.buf(gamma, (Cin,))
        db, bs_ = sinks.buf(beta, (Cin,))
        done = False
        _note(f"ln-bwd {M}x{Cin}", nbytes=5.0 * M * Cin * x.element_size())
        if gs and bs_ and ln_deferred.active():
            nblk = int(K.lib.lavt_layernorm_bwd_blocks(K.dt(dtype), M, Cin))
            wsd = ln_deferred.alloc(nblk * 2 * Cin, dev)
            if wsd is not None:
                job = dtable_chain.take_for_layernorm()
                rc = 1
                if job is not None:          # the binning of the attention backward issued a few launches ago rides in this launch
                    rc = K.lib.lavt_layernorm_bwd_partial_xn_dtable(K.dt(dtype), K.ptr(dxn), K.ptr(x), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), K.ptr(st[0]), K.ptr(st[1]),
                                                                    K.ptr(dx), K.ptr(xn), K.ptr(wsd), wsd.numel(), K.ptr(dy), M, Cin, C.byref(job), K.stream())
                    if rc == 0:
                        dtable_chain.done_by_layernorm()
                    elif rc != 1:
                        K.check(rc)
                if rc == 1:
                    K.check(K.lib.lavt_layernorm_bwd_partial_xn(K.dt(dtype), K.ptr(dxn), K.ptr(x), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), K.ptr(st[0]), K.ptr(st[1]), K.ptr(dx),
                                                                K.ptr(xn), K.ptr(wsd), wsd.numel(), K.ptr(dy), M, Cin, K.stream()))
                ln_deferred.add(wsd, nblk, Cin, dg, db, (gamma, beta))
                g_g = g_be = None
                done = True
        if not done:
            wsl = _scratch(int(K.lib.lavt_layernorm_bwd_blocks(K.dt(dtype), M, Cin)) * 2 * Cin, dev)
            K.check(K.lib.lavt_layernorm_bwd_xn(K.dt(dtype), K.ptr(dxn), K.ptr(x), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), K.ptr(st[0]), K.ptr(st[1]), K.ptr(dx), K.ptr(xn),
                                                K.ptr(dg), K.ptr(db), K.ptr(wsl), wsl.numel(), K.ptr(dy), M, Cin, K.stream()))
            g_g, g_be = sinks.done(gamma, dg, gs), sinks.done(beta, db, bs_)
        binary = o.row_scale is not None and o.row_scale_value != 0.0
        grads = []
        for (w, b, g, inp, n, kd, rs) in ((w2, b2, dy, h, Cout, Hd, True), (w1, b1, dpre, xn, Hd, Cin, False)):
            wbuf, wsink = sinks.buf(w, (n, kd))
            bbuf, bsink = sinks.buf(b, (n,)) if b is not None else (None, True)
            kw = dict(a_rowscale=o.row_scale, a_rowscale_div=o.row_scale_div, a_rowscale_binary=binary, alpha=o.row_scale_value if binary else 1.0) if rs and o.row_scale is not None else {}
            if wgrads.active() and wsink and bsink and (not kw or binary):
                gemm_tn(dtype, n, kd, M, g, n, inp, kd, wbuf, kd, colsum=bbuf, defer=wgrads, **kw)
                wgrads.notify(w)
                if b is not None:
                    wgrads.notify(b)
                grads += [None, None]
            else:
                gemm_tn(dtype, n, kd, M, g, n, inp, kd, wbuf, kd, colsum=bbuf, **kw)
                grads += [sinks.done(w, wbuf, wsink), sinks.done(b, bbuf, bsink) if b is not None else None]
        dw2, db2, dw1, db1 = grads
        return dx, g_g, g_be, dw1, db1, dw2, db2, None, None


def ln_mlp_ok(x, w1, b1, w2):
    """norm2 folded into fc1: bf16, biases present, widths multiples of 64 (the LDS-DMA GEMM's K tile) and the normalised width <= 1024"""
    return (x.dtype == torch.bfloat16 and b1 is not None and x.shape[1] % 64 == 0 and x.shape[1] <= 1024 and w1.shape[0] % 64 == 0 and w2.shape[0] % 64 == 0
            and os.environ.get("LAVT_LN_FOLD", "1") != "0" and os.environ.get("LAVT_FUSED_MLP", "1") != "0")


def ln_mlp(x, norm, w1, b1, w2, b2, **kw):
    """x + fc2(GELU(fc1(LN(x)))) * row_scale with the LayerNorm folded into fc1 (x is both the norm's input and the residual)"""
    return _LnMlp.apply(x, norm.weight, norm.bias, w1, b1, w2, b2, norm.eps, LinOpts(**kw))


def mlp(x, w1, b1, w2, b2, residual=None, **kw):
    """fc2(GELU(fc1(x))) [* row_scale] [+ residual]; the fused single-node form on the bf16 path, two `linear` ops otherwise"""
    o = LinOpts(**kw)
    # Measured on MI355X (Swin-B 2x480x480 step, same box, two repeats each).  With the library erff and 8-byte epilogue stores the fused node lost
    # (12.59 / 12.58 ms vs 12.51 / 12.53 ms); with the fast erf and the 16-byte paired-fragment stores it is level or slightly ahead
    # (11.94 / 11.91 ms vs 11.94 / 11.97 ms) and saves the [M, 4C] round trip.  LAVT_FUSED_MLP=0 selects the two-op form.
    if x.dtype == torch.bfloat16 and w1.shape[1] % 8 == 0 and w1.shape[0] % 64 == 0 and w2.shape[0] % 64 == 0 and os.environ.get("LAVT_FUSED_MLP", "1") != "0":
        return _Mlp.apply(x, w1, b1, w2, b2, residual, o)
    h = linear(x, w1, b1, act=K.ACT_GELU)
    return linear(h, w2, b2, residual=residual, **kw)


# ------------------------------------------------------------------------------------------ LayerNorm
@K.scoped
class _LayerNorm(torch.autograd.Function):
    """y = LN(x) (optionally over the PatchMerging 2x2 gather).  With `passthrough` the function also returns x itself: use that alias
    for the residual branch (x + f(LN(x))) and the two gradients of x meet inside the LayerNorm backward kernel (dx = LN'(dy) + dres)
    instead of in a separate element-wise add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, gather, rows, C, eps, passthrough):
        x = x.contiguous()
        y = torch.empty(rows, C, dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _note(f"ln {rows}x{C}", nbytes=2.0 * rows * C * x.element_size())
        K.check(K.lib.lavt_layernorm_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(gather), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), K.ptr(y),
                                         K.ptr(mean), K.ptr(rstd), rows, C, eps, K.stream()))
        ctx.save_for_backward(x, gamma, mean, rstd, gather, beta)
        ctx.rows, ctx.C, ctx.passthrough = rows, C, passthrough
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, mean, rstd, gather, beta = ctx.saved_tensors
        dy = dy.contiguous()
        if dres is not None:
            dres = dres.contiguous()
        dx = torch.empty_like(x)
        dg, gs = sinks.buf(gamma, (ctx.C,))
        db, bs = sinks.buf(beta, (ctx.C,))
        if gs and bs and ln_deferred.active():
            nblk = int(K.lib.lavt_layernorm_bwd_blocks(K.dt(x.dtype), ctx.rows, ctx.C))
            wsd = ln_deferred.alloc(nblk * 2 * ctx.C, x.device)
            if wsd is not None:
                _note(f"ln-bwd {ctx.rows}x{ctx.C}", nbytes=(4.0 if dres is not None else 3.0) * ctx.rows * ctx.C * x.element_size())
                job = dtable_chain.take_for_layernorm() if gather is None else None
                rc = 1
                if job is not None:          # a pending table-gradient binning job rides in this launch (the last one of a backward pass: the patch embedding's norm)
                    rc = K.lib.lavt_layernorm_bwd_partial_xn_dtable(K.dt(x.dtype), K.ptr(dy), K.ptr(x), K.ptr(_f32(gamma)), None, K.ptr(mean), K.ptr(rstd), K.ptr(dx), None,
                                                                    K.ptr(wsd), wsd.numel(), K.ptr(dres), ctx.rows, ctx.C, C.byref(job), K.stream())
                    if rc == 0:
                        dtable_chain.done_by_layernorm()
                    elif rc != 1:
                        K.check(rc)
                if rc == 1:
                    K.check(K.lib.lavt_layernorm_bwd_partial(K.dt(x.dtype), K.ptr(dy), K.ptr(x), K.ptr(gather), K.ptr(_f32(gamma)), K.ptr(mean), K.ptr(rstd),
                                                             K.ptr(dx), K.ptr(wsd), wsd.numel(), K.ptr(dres), ctx.rows, ctx.C, K.stream()))
                ln_deferred.add(wsd, nblk, ctx.C, dg, db, (gamma, beta))
                return dx, None, None, None, None, None, None, None
        ws = _scratch(int(K.lib.lavt_layernorm_bwd_blocks(K.dt(x.dtype), ctx.rows, ctx.C)) * 2 * ctx.C, x.device)     # (without it the kernel falls back to same-address atomics)
        K.check(K.lib.lavt_layernorm_bwd(K.dt(x.dtype), K.ptr(dy), K.ptr(x), K.ptr(gather), K.ptr(_f32(gamma)), K.ptr(mean), K.ptr(rstd),
                                         K.ptr(dx), K.ptr(dg), K.ptr(db), K.ptr(ws), ws.numel(), K.ptr(dres), ctx.rows, ctx.C, K.stream()))
        return dx, sinks.done(gamma, dg, gs), sinks.done(beta, db, bs), None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, gather=None):
    """x [rows, C]; with gather (int32 [rows_out, 4]) the input row is the concat of 4 source rows of x (PatchMerging)."""
    if gather is None:
        return _LayerNorm.apply(x, gamma, beta, None, x.shape[0], x.shape[1], eps, False)
    return _LayerNorm.apply(x, gamma, beta, gather, gather.shape[0], 4 * x.shape[1], eps, False)


def layer_norm_res(x, gamma, beta, eps=1e-5):
    """-> (LN(x), x'): x' aliases x; feed it to the residual branch so that both gradients of x are summed inside the LN backward kernel"""
    return _LayerNorm.apply(x, gamma, beta, None, x.shape[0], x.shape[1], eps, True)


# ------------------------------------------------------------------------------------------ window attention core
def _win3(win):
    """window spec -> (wd, wh, ww): int ws = 2-D Swin (1, ws, ws); tuple = Video-Swin window"""
    return (1, win, win) if isinstance(win, int) else tuple(int(v) for v in win)


FUSED_ATTN_MAX_N = 160          # exact-fp32 fused kernels: one window's K/V (and P for backward) up to 12x12 = 144 (+pad) tokens
FUSED_ATTN_MAX_N_BF16 = 400     # bf16 MFMA kernels: Q/K/V(/dO) of a whole window in LDS up to 25 key tiles (Video-Swin 8x7x7 = 392 tokens)


@K.scoped
class _WindowAttn(torch.autograd.Function):
    """Fused kernels (N <= 160 tokens per window).  qkv [nwin*N, 3C] windowed rows."""

    @staticmethod
    def forward(ctx, qkv, table, region, win, heads, N):
        qkv = qkv.contiguous()
        wd, wh, ww = win
        C3 = qkv.shape[1]
        Cc = C3 // 3
        nwin = qkv.shape[0] // N
        dev = qkv.device
        ld = 64 if N <= 64 else (160 if N <= 160 else 416)   # padded key axis (pairs of 16-wide MFMA tiles); 416 = 13 pairs for 392-token video windows
        dense = None
        if not K.lib.lavt_attn_uses_table(K.dt(qkv.dtype), N):      # exact-fp32 kernels read the dense bias (padding columns hold -1e30)
            dense = torch.empty(heads, N, ld, dtype=torch.float32, device=dev)
            K.check(K.lib.lavt_relpos_expand(K.ptr(_f32(table)), K.ptr(dense), wd, wh, ww, N, heads, ld, K.stream()))
        out = torch.empty(nwin * N, Cc, dtype=qkv.dtype, device=dev)
        lse = torch.empty(nwin, heads, N, dtype=torch.float32, device=dev)
        nw_img = region.shape[0] if region is not None else 0
        scale = float((Cc // heads) ** -0.5)
        _note(f"wattn {nwin * N}x{Cc} N{N}", 4.0 * nwin * heads * N * N * 32)
        K.check(K.lib.lavt_window_attn_fwd(K.dt(qkv.dtype), K.ptr(qkv), K.ptr(dense), ld, K.ptr(region), nw_img, K.ptr(out), K.ptr(lse),
                                           K.ptr(_f32(table)), wd, wh, ww, nwin, N, heads, Cc // heads, scale, K.stream()))
        ctx.save_for_backward(qkv, dense, region, out, lse, table)
        ctx.dims = (win, heads, nwin, N, Cc, nw_img, scale, ld)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, dense, region, out, lse, table = ctx.saved_tensors
        win, heads, nwin, N, Cc, nw_img, scale, ld = ctx.dims
        wd, wh, ww = win
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        nws = int(K.lib.lavt_window_attn_bwd_ws(K.dt(qkv.dtype), nwin, N, heads, ld, wd, wh, ww))
        ws = torch.empty(nws, dtype=torch.float32, device=qkv.device) if nws > 0 else None                      # dS slabs + table histograms
        R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1)
        dtable, ts = sinks.buf(table, (R, heads))
        pieces = int(K.lib.lavt_window_attn_bwd_pieces(K.dt(qkv.dtype), nwin, N, heads, ld)) if (ts and ln_deferred.active()) else 0
        parts = ln_deferred.alloc(pieces * heads * R, qkv.device) if pieces > 0 else None
        _note(f"wattn-bwd {nwin * N}x{Cc} N{N}", 10.0 * nwin * heads * N * N * 32)
        if parts is not None and ws is not None and dtable_chain.enabled:
            dtable_chain.launch(qkv.dtype, qkv, ld, region, nw_img, out, dout, lse, dqkv, table, ws, parts, wd, wh, ww, nwin, N, heads, Cc // heads, scale)
        else:
            K.check(K.lib.lavt_window_attn_bwd(K.dt(qkv.dtype), K.ptr(qkv), K.ptr(dense), ld, K.ptr(region), nw_img, K.ptr(out), K.ptr(dout),
                                               K.ptr(lse), K.ptr(dqkv), K.ptr(_f32(table)), None if parts is not None else K.ptr(dtable), K.ptr(ws),
                                               ws.numel() if ws is not None else 0, K.ptr(parts), wd, wh, ww, nwin, N, heads, Cc // heads, scale, K.stream()))
        if parts is not None:          # table gradient finished with the other layers' in one launch at the end of backward
            ln_deferred.add_table(parts, pieces, heads, R, dtable, table)
            return dqkv, None, None, None, None, None
        return dqkv, sinks.done(table, dtable, ts), None, None, None, None


@K.scoped
class _WindowAttnComposed(torch.autograd.Function):
    """Windows too large for the fused kernels (Video-Swin: 8x7x7 = 392, 8x12x12 = 1152 tokens):
    S = scale q k^T (gather-GEMM batched over (window, head)) -> bias + shift mask + softmax (lavt_attn_softmax) -> P v (GEMM).
    qkv is first regrouped head-major ([nwin][heads][Np][q|k|v x 32]) so that every GEMM is ONE launch with a uniform batch stride.
    Scores are materialised ([nwin*heads, Np, Np]): correct for any N and both dtypes; a streaming (flash-style) MFMA kernel for these
    sizes is future work.  Windows whose token count is not a multiple of 8 are zero-padded per window."""

    @staticmethod
    def forward(ctx, qkv, table, region, win, heads, N):
        dtype, dev = qkv.dtype, qkv.device
        wd, wh, ww = win
        C3 = qkv.shape[1]
        Cc = C3 // 3
        nwin = qkv.shape[0] // N
        Np = -(-N // 8) * 8
        q5 = qkv.view(nwin, N, 3, heads, 32)
        if Np != N:
            q5 = torch.nn.functional.pad(q5, (0, 0, 0, 0, 0, 0, 0, Np - N))
        qh = q5.permute(0, 3, 1, 2, 4).contiguous().view(nwin * heads * Np, 96)            # rows (window, head, token); columns q | k | v
        nb = nwin * heads
        dense = torch.empty(heads, N, Np, dtype=torch.float32, device=dev)
        K.check(K.lib.lavt_relpos_expand(K.ptr(_f32(table)), K.ptr(dense), wd, wh, ww, N, heads, Np, K.stream()))
        nw_img = region.shape[0] if region is not None else 0
        scale = float((Cc // heads) ** -0.5)
        S = torch.empty(nb * Np, Np, dtype=dtype, device=dev)
        gemm_nt(dtype, Np, Np, 32, qh, 96, qh, 96, S, Np, batch=nb, strideA=Np * 96, strideB=Np * 96, strideC=Np * Np, alpha=scale, b_off=32)
        P = torch.empty_like(S)
        K.check(K.lib.lavt_attn_softmax_fwd(K.dt(dtype), K.ptr(S), K.ptr(dense), Np, K.ptr(region), nw_img, K.ptr(P), nb * Np, Np, N, Np, heads, K.stream()))
        del S
        oh = torch.empty(nb * Np, 32, dtype=dtype, device=dev)
        gemm_nt(dtype, Np, 32, Np, P, Np, qh, 96, oh, 32, batch=nb, strideA=Np * Np, strideB=Np * 96, strideC=Np * 32, b_kmajor=True, b_off=64)
        out = oh.view(nwin, heads, Np, 32)[:, :, :N].permute(0, 2, 1, 3).reshape(nwin * N, Cc)
        ctx.save_for_backward(qh, dense, region, table, P)
        ctx.dims = (win, heads, nwin, N, Np, Cc, nw_img, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qh, dense, region, table, P = ctx.saved_tensors
        win, heads, nwin, N, Np, Cc, nw_img, scale = ctx.dims
        wd, wh, ww = win
        dtype, dev = qh.dtype, qh.device
        nb = nwin * heads
        d4 = dout.reshape(nwin, N, heads, 32)
        if Np != N:
            d4 = torch.nn.functional.pad(d4, (0, 0, 0, 0, 0, Np - N))
        doh = d4.permute(0, 2, 1, 3).contiguous().view(nb * Np, 32)
        dS = torch.empty_like(P)                 # first dP = dO V^T, then overwritten with dS
        gemm_nt(dtype, Np, Np, 32, doh, 32, qh, 96, dS, Np, batch=nb, strideA=Np * 32, strideB=Np * 96, strideC=Np * Np, b_off=64)
        K.check(K.lib.lavt_attn_softmax_bwd(K.dt(dtype), K.ptr(P), K.ptr(dS), nb * Np, N, Np, K.stream()))
        ddense = torch.empty(heads, N, Np, dtype=torch.float32, device=dev)      # bias gradient: sum of dS over the windows
        K.check(K.lib.lavt_attn_dbias_sum(K.dt(dtype), K.ptr(dS), K.ptr(ddense), nwin, heads, N, Np, Np, K.stream()))
        dqh = torch.empty(nb * Np, 96, dtype=dtype, device=dev)
        # dQ = scale dS K
        gemm_nt(dtype, Np, 32, Np, dS, Np, qh, 96, dqh, 96, batch=nb, strideA=Np * Np, strideB=Np * 96, strideC=Np * 96, b_kmajor=True, alpha=scale, b_off=32)
        # dK = scale dS^T Q ; dV = P^T dO   (fp32 outputs of the wgrad family, then cast into the head-major gradient)
        dkv = torch.zeros(2, nb * Np, 32, dtype=torch.float32, device=dev)
        gemm_tn(dtype, Np, 32, Np, dS, Np, qh, 96, dkv[0], 32, batch=nb, strideA=Np * Np, strideB=Np * 96, strideC=Np * 32, alpha=scale)
        gemm_tn(dtype, Np, 32, Np, P, Np, doh, 32, dkv[1], 32, batch=nb, strideA=Np * Np, strideB=Np * 32, strideC=Np * 32)
        dqh.view(nb * Np, 3, 32)[:, 1:] = dkv.permute(1, 0, 2).to(dtype)
        dqkv = dqh.view(nwin, heads, Np, 3, 32)[:, :, :N].permute(0, 2, 3, 1, 4).reshape(nwin * N, 3 * Cc)
        dtable, ts = sinks.buf(table, ((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), heads))
        K.check(K.lib.lavt_relpos_reduce(K.ptr(ddense), K.ptr(dtable), wd, wh, ww, N, heads, Np, K.stream()))
        return dqkv, sinks.done(table, dtable, ts), None, None, None, None


@K.scoped
class _WmsaFused(torch.autograd.Function):
    """norm1 -> window partition / shift / pad -> qkv -> attention core of a Swin block as ONE forward kernel (csrc/wmsa_fused.hip; reference
    lib/backbone.py:201-217 + 113-140).  x [tokens, C] is the residual stream; returns (o [nwin * N, C] in window order -- the proj GEMM scatters it
    back and adds the residual --, x' aliasing x for the residual branch).  The kernel also leaves qkv, the LayerNorm output and its row statistics
    for the backward pass, which is the unfused sequence: attention backward -> qkv data / weight gradients -> LayerNorm backward (+ residual gradient)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, wq, bq, table, region, wmap, ws, heads, eps, tok=None):
        x = x.contiguous()
        dtype, dev = x.dtype, x.device
        M, Cc = x.shape
        N = ws * ws
        Mw = wmap.numel()
        nwin = Mw // N
        Wg, wsum, biasp = weights.get_lnfold(wq, bq, gamma, beta)
        out = torch.empty(Mw, Cc, dtype=dtype, device=dev)
        qkv = torch.empty(Mw, 3 * Cc, dtype=dtype, device=dev)
        xn = torch.empty_like(x)
        lse = torch.empty(nwin, heads, N, dtype=torch.float32, device=dev)
        st = torch.empty(2, M, dtype=torch.float32, device=dev)
        nw_img = region.shape[0] if region is not None else 0
        scale = float((Cc // heads) ** -0.5)
        _note(f"wmsa {Mw}x{Cc} N{N}", 2.0 * Mw * Cc * 3 * Cc + 4.0 * nwin * heads * N * N * 32)
        # (a launch that leaves resident slots free -- <= 400 (window, head) workgroups for 512 -- also zeroes a slice of the step's gradient buffer)
        fptr, fbytes = fill_riders.take() if nwin * heads <= 400 else (None, 0)
        K.check(K.lib.lavt_wmsa_fwd_rider(K.ptr(x), K.ptr(wmap), K.ptr(Wg), K.ptr(wsum), K.ptr(biasp), K.ptr(_f32(bq)), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), K.ptr(_f32(table)),
                                          K.ptr(region), nw_img, K.ptr(out), K.ptr(lse), K.ptr(qkv), K.ptr(xn), K.ptr(st[0]), K.ptr(st[1]), _zero_page(dev), ws, nwin, N,
                                          heads, Cc, eps, scale, fptr, fbytes, K.stream()))
        ctx.save_for_backward(x, gamma, beta, wq, bq, table, region, wmap, out, qkv, xn, lse, st)
        ctx.dims = (ws, heads, nwin, N, Cc, nw_img, scale, Mw)
        ctx.tok = tok if tok is not None else (None, None)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, dout, dres):
        x, gamma, beta, wq, bq, table, region, wmap, out, qkv, xn, lse, st = ctx.saved_tensors
        ws_, heads, nwin, N, Cc, nw_img, scale, Mw = ctx.dims
        dtype, dev = x.dtype, x.device
        M = x.shape[0]
        ld = 64 if N <= 64 else 160
        # ---- attention core (as _WindowAttn.backward) ----
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        nws = int(K.lib.lavt_window_attn_bwd_ws(K.dt(dtype), nwin, N, heads, ld, 1, ws_, ws_))
        wsb = torch.empty(nws, dtype=torch.float32, device=dev) if nws > 0 else None
        R = (2 * ws_ - 1) * (2 * ws_ - 1)
        dtable, ts = sinks.buf(table, (R, heads))
        pieces = int(K.lib.lavt_window_attn_bwd_pieces(K.dt(dtype), nwin, N, heads, ld)) if (ts and ln_deferred.active()) else 0
        parts = ln_deferred.alloc(pieces * heads * R, dev) if pieces > 0 else None
        _note(f"wattn-bwd {Mw}x{Cc} N{N}", 10.0 * nwin * heads * N * N * 32)
        if parts is not None and wsb is not None and dtable_chain.enabled:
            dtable_chain.launch(dtype, qkv, ld, region, nw_img, out, dout, lse, dqkv, table, wsb, parts, 1, ws_, ws_, nwin, N, heads, Cc // heads, scale)
        else:
            K.check(K.lib.lavt_window_attn_bwd(K.dt(dtype), K.ptr(qkv), None, ld, K.ptr(region), nw_img, K.ptr(out), K.ptr(dout), K.ptr(lse), K.ptr(dqkv), K.ptr(_f32(table)),
                                               None if parts is not None else K.ptr(dtable), K.ptr(wsb), wsb.numel() if wsb is not None else 0, K.ptr(parts), 1, ws_, ws_,
                                               nwin, N, heads, Cc // heads, scale, K.stream()))
        if parts is not None:
            ln_deferred.add_table(parts, pieces, heads, R, dtable, table)
            g_table = None
        else:
            g_table = sinks.done(table, dtable, ts)
        # ---- qkv projection: data gradient scattered back to token order, weight / bias gradient over the windowed rows ----
        Wc = weights.get(wq, dtype, "lin")
        dxn = torch.empty_like(x)
        inv, padrows = ctx.tok
        if inv is not None and os.environ.get("LAVT_TOKEN_ORDER_DGRAD", "1") != "0":
            # token order: one output row per real token, its dqkv row gathered through the inverse window map (the rows of padded window
            # positions are never computed: 1800 instead of 2592 rows at stage 2, 450 instead of 1152 at stage 3)
            gemm_nt(dtype, M, Cc, 3 * Cc, dqkv, 3 * Cc, Wc, Cc, dxn, Cc, b_kmajor=True, a_rowmap=inv)
        else:
            gemm_nt(dtype, Mw, Cc, 3 * Cc, dqkv, 3 * Cc, Wc, Cc, dxn, Cc, b_kmajor=True, c_rowmap=wmap)
        # ---- LayerNorm backward (the residual branch's gradient joins inside the kernel, as _LayerNorm.backward), prepared FIRST: under the step harness its
        # launch rides on the block's grouped weight-gradient launch, which the qkv member below completes (ops._WgradQueue.add(rider=...))
        if dres is not None:
            dres = dres.contiguous()
        dx = torch.empty_like(x)
        dg, gs = sinks.buf(gamma, (Cc,))
        db, bs_ = sinks.buf(beta, (Cc,))
        rider, nblk, wsd = None, 0, None
        if gs and bs_ and ln_deferred.active():
            nblk = int(K.lib.lavt_layernorm_bwd_blocks(K.dt(dtype), M, Cc))
            wsd = ln_deferred.alloc(nblk * 2 * Cc, dev)
            if wsd is not None:
                rider = (dxn, x, _f32(gamma), st[0], st[1], dx, wsd, dres, M, Cc)
        # ---- qkv weight / bias gradient ----
        wbuf, wsink = sinks.buf(wq, (3 * Cc, Cc))
        bbuf, bsink = sinks.buf(bq, (3 * Cc,))
        rode = False
        if wgrads.active() and wsink and bsink and inv is not None and padrows.numel() > 0 and os.environ.get("LAVT_TOKEN_ORDER_WGRAD", "1") != "0":
            # token order: dW = sum over the REAL tokens of dqkv[inv[t]]^T xn[t] (padded window positions have xn = 0: K = tokens instead of window
            # rows); their dq / dk / dv still belong to the bias gradient (the reference pads after norm1): a side member of the grouped launch sums
            # those rows alone (B = the zero page, 8 dummy columns), both column sums added atomically into the zeroed bias gradient
            dummy = _discard_out(3 * Cc * 8, dev)
            zp = _zero_page_tensor(dev)
            wgrads.notify(wq)          # (before the member is queued: it closes the group, and the flush reports what has been notified)
            wgrads.notify(bq)
            gemm_tn(dtype, 3 * Cc, 8, padrows.numel(), dqkv, 3 * Cc, zp, 0, dummy, 8, a_rowmap=padrows, colsum=bbuf, colsum_atomic=True, defer=wgrads, extra=True)
            gemm_tn(dtype, 3 * Cc, Cc, M, dqkv, 3 * Cc, xn, Cc, wbuf, Cc, a_rowmap=inv, colsum=bbuf, colsum_atomic=True, defer=wgrads, rider=rider)
            rode = rider is not None
            g_w = g_b = None
        elif wgrads.active() and wsink and bsink:
            wgrads.notify(wq)
            wgrads.notify(bq)
            gemm_tn(dtype, 3 * Cc, Cc, Mw, dqkv, 3 * Cc, xn, Cc, wbuf, Cc, b_rowmap=wmap, colsum=bbuf, defer=wgrads, rider=rider)
            rode = rider is not None
            g_w = g_b = None
        else:
            gemm_tn(dtype, 3 * Cc, Cc, Mw, dqkv, 3 * Cc, xn, Cc, wbuf, Cc, b_rowmap=wmap, colsum=bbuf)
            g_w, g_b = sinks.done(wq, wbuf, wsink), sinks.done(bq, bbuf, bsink)
        if rider is not None:
            if not rode:
                _note(f"ln-bwd {M}x{Cc}", nbytes=(4.0 if dres is not None else 3.0) * M * Cc * x.element_size())
                _launch_ln_partial(rider)
            ln_deferred.add(wsd, nblk, Cc, dg, db, (gamma, beta))
            g_g = g_be = None
        else:
            wsl = _scratch(int(K.lib.lavt_layernorm_bwd_blocks(K.dt(dtype), M, Cc)) * 2 * Cc, dev)
            K.check(K.lib.lavt_layernorm_bwd(K.dt(dtype), K.ptr(dxn), K.ptr(x), None, K.ptr(_f32(gamma)), K.ptr(st[0]), K.ptr(st[1]), K.ptr(dx), K.ptr(dg), K.ptr(db),
                                             K.ptr(wsl), wsl.numel(), K.ptr(dres), M, Cc, K.stream()))
            g_g, g_be = sinks.done(gamma, dg, gs), sinks.done(beta, db, bs_)
        return dx, g_g, g_be, g_w, g_b, g_table, None, None, None, None, None, None


_WMSA_FUSED_MAX_C = int(os.environ.get("LAVT_WMSA_FUSED_MAX_C", "1024"))


def wmsa_fused_ok(x, ws, heads, has_bias):
    """the one-kernel W-MSA forward covers bf16 2-D windows of <= 160 tokens with C = 32 heads a multiple of 64 and a qkv bias.  Measured on MI355X
    (tools/wmsa_time.py, Swin-B w12 stage shapes at batch 2, norm1 + qkv + attention, graph-replayed): C = 128: 38.5 vs 45.2 us unfused, 256: 23.4 vs
    38.0, 512: 25.6-26.8 vs 30.5-31.7, 1024: 29.7 vs 31.6 (33.6 before the row-statistics reads of the K loop went through inline asm: hipcc put a
    vmcnt(0) in front of the plain LDS load, which serialised the ring)."""
    return (x.dtype == torch.bfloat16 and has_bias and x.shape[1] == 32 * heads and x.shape[1] % 64 == 0 and x.shape[1] <= _WMSA_FUSED_MAX_C
            and ws * ws <= 160 and os.environ.get("LAVT_WMSA_FUSED", "1") != "0")


def wmsa_fused(x, norm, attn, region, wmap, ws, heads, tok=None):
    """x [tokens, C], norm = the block's norm1, attn = its WindowAttention (parameter containers) -> (attention output in window order, x');
    tok = (inverse row map, padded-row list) of rowmaps for the token-order weight gradient"""
    return _WmsaFused.apply(x, norm.weight, norm.bias, attn.qkv.weight, attn.qkv.bias, attn.relative_position_bias_table, region, wmap, ws, heads, norm.eps, tok)


def window_attention(qkv, table, region, win, heads, N=None):
    """win: int ws (2-D) or (wd, wh, ww); N: tokens per window (default: the full window; smaller for clipped video windows)."""
    win = _win3(win)
    if N is None:
        N = win[0] * win[1] * win[2]
    if N <= FUSED_ATTN_MAX_N or (qkv.dtype == torch.bfloat16 and N <= FUSED_ATTN_MAX_N_BF16 and K.lib.lavt_attn_uses_table(K.dt(qkv.dtype), N)
                                 and os.environ.get("LAVT_ATTN_COMPOSED", "0") != "1"):
        return _WindowAttn.apply(qkv, table, region, win, heads, N)
    return _WindowAttnComposed.apply(qkv, table, region, win, heads, N)


# ------------------------------------------------------------------------------------------ Instance / Batch norm
def _stats(x, groups, rows, Cc):
    s = torch.empty(2, groups, Cc, dtype=torch.float32, device=x.device)       # cleared by the kernel (two-stage form)
    ws = _scratch(1025 * groups * 2 * Cc, x.device)
    K.check(K.lib.lavt_colstats(K.dt(x.dtype), K.ptr(x), K.ptr(s[0]), K.ptr(s[1]), K.ptr(ws), ws.numel(), groups, rows, Cc, K.stream()))
    return s


@K.scoped
class _InstanceNorm(torch.autograd.Function):
    """y = IN_over_rows(x) (* mul); x [B*T, C], statistics per (b, c) over the T rows (lib/backbone.py:1311-1327)."""

    @staticmethod
    def forward(ctx, x, mul, B, T):
        x = x.contiguous()
        Cc = x.shape[1]
        mean = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = _scratch(1025 * B * 2 * Cc, x.device)
        K.check(K.lib.lavt_colstats_meanrstd(K.dt(x.dtype), K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(ws), ws.numel(), B, T, Cc, 1e-5, None, None, 0.0, K.stream()))
        if mul is not None:
            mul = mul.contiguous()
        y = torch.empty_like(x)
        K.check(K.lib.lavt_norm_apply(K.dt(x.dtype), K.ptr(x), K.ptr(mean), K.ptr(rstd), None, None, K.ptr(mul), 0, K.ptr(y), B, T, Cc, K.stream()))
        ctx.save_for_backward(x, mul, mean, rstd)
        ctx.dims = (B, T, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mul, mean, rstd = ctx.saved_tensors
        B, T, Cc = ctx.dims
        dy = dy.contiguous()
        s = torch.empty(2, B, Cc, dtype=torch.float32, device=x.device)         # cleared by lavt_norm_bwd_stats (two-stage form)
        d = K.dt(x.dtype)
        ws = _scratch(1025 * B * 2 * Cc, x.device)
        K.check(K.lib.lavt_norm_bwd_stats(d, K.ptr(dy), K.ptr(x), None, K.ptr(mean), K.ptr(rstd), None, None, K.ptr(mul), 0,
                                          K.ptr(s[0]), K.ptr(s[1]), K.ptr(ws), ws.numel(), B, T, Cc, K.stream()))
        dx = torch.empty_like(x)
        dmul = torch.empty_like(x) if mul is not None else None
        K.check(K.lib.lavt_norm_bwd_apply(d, K.ptr(dy), K.ptr(x), None, K.ptr(mean), K.ptr(rstd), None, None, K.ptr(mul), 0,
                                          K.ptr(s[0]), K.ptr(s[1]), float(T), K.ptr(dx), K.ptr(dmul), B, T, Cc, K.stream()))
        return dx, dmul, None, None


def instance_norm(x, B, T, mul=None):
    return _InstanceNorm.apply(x, mul, B, T)


def combine_rank_stats(allst: torch.Tensor, rows_per_rank: int) -> torch.Tensor:
    """allst [world, 2, ...]: every rank's (sum x, centred second moment M2 = sum (x - mean_rank)^2) over its `rows_per_rank` rows
    -> [2, ...] the same two quantities over all ranks' rows (parallel-variance combination; no E[x^2] - E[x]^2 anywhere)."""
    world = allst.shape[0]
    tot = allst[:, 0].sum(0)
    gmean = tot / float(rows_per_rank * world)
    m2 = allst[:, 1].sum(0) + rows_per_rank * ((allst[:, 0] / rows_per_rank - gmean) ** 2).sum(0)
    return torch.stack([tot, m2])


def syncbn_gather(s: torch.Tensor, group) -> torch.Tensor:
    """ONE collective: every rank's (sum x, centred second moment) pair -> [world, 2, 1, C]"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = torch.empty(world * s.numel(), dtype=torch.float32, device=s.device)          # flat in / flat out: the layout every backend accepts
    dist.all_gather_into_tensor(flat, s.contiguous().view(-1), group=group)
    return flat.view((world,) + tuple(s.shape))


def syncbn_exchange_forward(s: torch.Tensor, rows: int, group):
    """SyncBatchNorm forward exchange (train.py:589 semantics): `s` [2, 1, C] = this rank's (sum x, centred second moment) over its `rows` rows.
    ONE collective gathers every rank's pair; the pairs are combined locally (Chan et al.; equal row counts per rank, as DistributedSampler with
    drop_last gives).  -> (statistics of the global batch [2, 1, C], global row count)"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = torch.empty(world * s.numel(), dtype=torch.float32, device=s.device)          # flat in / flat out: the layout every backend accepts
    dist.all_gather_into_tensor(flat, s.contiguous().view(-1), group=group)
    return combine_rank_stats(flat.view((world,) + tuple(s.shape)), rows), float(rows * world)


def syncbn_exchange_backward(s: torch.Tensor, group):
    """SyncBatchNorm backward exchange: `s` [2, C] = this rank's (sum dy, sum dy * xhat); summed over the ranks in place (one all-reduce)"""
    import torch.distributed as dist
    dist.all_reduce(s, group=group)
    return s


_CONV_STATS = os.environ.get("LAVT_CONV_STATS", "1") != "0"


class _ConvStats:
    """Column statistics that a convolution's epilogue stored beside its output (csrc/gemm_nt_pipe.hip: per block of rows the column sums and the second
    moments about the block's mean, from the fp32 accumulators): the BatchNorm that follows (reference lib/mask_predictor.py:60-97) combines the blocks
    instead of reading the [M, C] map again.  Keyed by the output's storage; one entry per device is enough (the consumer is the next op)."""

    def __init__(self):
        self.ent = {}
        self.hits = 0

    def put(self, y, st):
        self.ent[y.device] = (weakref.ref(y), y.data_ptr(), tuple(y.shape), st)

    def take(self, x):
        """the statistics of x if x IS the tensor the last convolution on its device returned (same object, same storage): a tensor that merely
        reuses a freed output's address does not qualify"""
        e = self.ent.pop(x.device, None)
        if e is not None and e[0]() is x and e[1] == x.data_ptr() and e[2] == tuple(x.shape):
            self.hits += 1
            return e[3]
        return None


conv_stats = _Proxy("conv_stats")


class StepContext:
    """the state the ops keep between calls (see the note at the top of this module).  `with use_context(ctx):` makes it current.

    The current context is ONE pointer per process, not per thread, on purpose: autograd runs the backward of GPU ops on its own worker thread, which has to
    see the context of the thread that called `backward()` inside the `with` block (and that thread is blocked meanwhile).  Harnesses in private contexts
    can therefore alternate freely, but two host threads must not run forward / backward passes at the same time."""

    def __init__(self):
        self.weights = _WeightCache()
        self.fp8 = _Fp8State()
        self.sinks = _GradSinks()
        self.zero_arena = _ZeroArena()
        self.fill_riders = _FillRiders()
        self.wgrads = _WgradQueue()
        self.ln_deferred = _LnDeferred()
        self.dtable_chain = _DtableChain()
        self.conv_stats = _ConvStats()
        self.tn_parts, self.sink_out, self.sk_scratch = {}, {}, {}          # per-device scratch of the split reductions (single-stream order within a context)
        self.dp_state = {}                                                   # per-device DropPath generator (seed, draw counter)

    def __enter__(self):
        global _ctx
        self._prev = getattr(self, "_prev", [])
        self._prev.append(_ctx)
        _ctx = self
        return self

    def __exit__(self, *exc):
        global _ctx
        _ctx = self._prev.pop()
        return False


_default_ctx = StepContext()
_ctx = _default_ctx


def default_context():
    return _default_ctx


def current_context():
    return _ctx


def use_context(ctx):
    """`with use_context(ctx):` -- ops called inside (forward, the autograd backward started inside, optimizer refreshes) use ctx's state"""
    return ctx if ctx is not None else _default_ctx


class _HipBnKernels:
    """The local passes of BatchNorm + ReLU on NHWC rows (csrc/norm.hip).  _BatchNormRelu talks to them through this small interface so that the
    multi-rank protocol around them (what is exchanged, when, with which counts) can be driven by a CPU stand-in in the gloo tests."""
    fused_sinks = True          # bwd_stats can accumulate the local sums straight into the parameters' gradient sinks (the shipping backward branch)

    @staticmethod
    def stats(x):
        st = conv_stats.take(x)
        if st is not None:              # the producing convolution left block statistics: combine them (no pass over x)
            parts, nblk, rpb = st
            s = torch.empty(2, 1, x.shape[1], dtype=torch.float32, device=x.device)
            K.check(K.lib.lavt_colstats_finish_blocks(K.ptr(parts), nblk, rpb, x.shape[0], x.shape[1], 0.0, None, None, K.ptr(s[0]), K.ptr(s[1]), None, None, 0.0, K.stream()))
            return s
        return _stats(x, 1, x.shape[0], x.shape[1])                      # [2, 1, C]: sum, centred second moment of the local rows

    @staticmethod
    def stats_fused(x, eps, running_mean, running_var, momentum):
        """single-rank training statistics: sums -> mean / rstd (+ running estimates) in two launches (lavt_colstats_meanrstd)"""
        R, Cc = x.shape
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        st = conv_stats.take(x)
        if st is not None:              # the producing convolution left block statistics: one small launch, no pass over x
            parts, nblk, rpb = st
            K.check(K.lib.lavt_colstats_finish_blocks(K.ptr(parts), nblk, rpb, R, Cc, eps, K.ptr(mean), K.ptr(rstd), None, None, K.ptr(running_mean),
                                                      K.ptr(running_var), momentum, K.stream()))
            return mean, rstd
        ws = _scratch(1025 * 2 * Cc, x.device)
        K.check(K.lib.lavt_colstats_meanrstd(K.dt(x.dtype), K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(ws), ws.numel(), 1, R, Cc, eps, K.ptr(running_mean),
                                             K.ptr(running_var), momentum, K.stream()))
        return mean, rstd

    @staticmethod
    def combine_finalize(allst, rows, eps, running_mean, running_var, momentum):
        """allst [world, 2, 1, C] (every rank's sum / centred M2 over `rows` rows) -> mean, rstd of the global batch; running estimates updated"""
        world, Cc = allst.shape[0], allst.shape[-1]
        mean = torch.empty(Cc, dtype=torch.float32, device=allst.device)
        rstd = torch.empty_like(mean)
        K.check(K.lib.lavt_syncbn_combine(K.ptr(allst), world, float(rows), eps, K.ptr(mean), K.ptr(rstd), K.ptr(running_mean), K.ptr(running_var),
                                          momentum, Cc, K.stream()))
        return mean, rstd

    @staticmethod
    def finalize(s, count, eps, running_mean, running_var, momentum):
        Cc = s.shape[-1]
        mean = torch.empty(Cc, dtype=torch.float32, device=s.device)
        rstd = torch.empty_like(mean)
        K.check(K.lib.lavt_stats_finalize(K.ptr(s[0]), K.ptr(s[1]), count, eps, K.ptr(mean), K.ptr(rstd), K.ptr(running_mean), K.ptr(running_var),
                                          momentum, Cc, K.stream()))
        return mean, rstd

    @staticmethod
    def apply(x, mean, rstd, gamma, beta, fp8_site=None):
        y = torch.empty_like(x)
        if fp8_site is not None and x.dtype == torch.bfloat16 and fp8_enabled() and fp8.step_active:
            q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            a_prev, a_cur = fp8.site_ptrs(fp8_site, x.device)
            K.check(K.lib.lavt_norm_apply_q8(K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), None, 1, K.ptr(y), K.ptr(q), a_prev, a_cur,
                                             1, x.shape[0], x.shape[1], K.stream()))
            fp8.put_twin(y, q, a_prev)
            return y
        K.check(K.lib.lavt_norm_apply(K.dt(x.dtype), K.ptr(x), K.ptr(mean), K.ptr(rstd), K.ptr(_f32(gamma)), K.ptr(_f32(beta)), None, 1,
                                      K.ptr(y), 1, x.shape[0], x.shape[1], K.stream()))
        return y

    @staticmethod
    def bwd_stats(dy, x, y, mean, rstd, gamma, beta, out=None):
        """-> (sum dy', sum dy' * xhat) as two [C] fp32 tensors; `out` = two zeroed buffers to accumulate into (the parameters' gradient sinks)"""
        R, Cc = x.shape
        if out is None:
            s = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
            out = (s[0], s[1])
        ws = _scratch(1025 * 2 * Cc, x.device)
        K.check(K.lib.lavt_norm_bwd_stats(K.dt(x.dtype), K.ptr(dy), K.ptr(x), K.ptr(y), K.ptr(mean), K.ptr(rstd), K.ptr(_f32(gamma)), K.ptr(_f32(beta)),
                                          None, 1, K.ptr(out[0]), K.ptr(out[1]), K.ptr(ws), ws.numel(), 1, R, Cc, K.stream()))
        return out

    @staticmethod
    def bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count, fp8_dy_site=None):
        R, Cc = x.shape
        dx = torch.empty_like(x)
        if fp8_dy_site is not None and x.dtype == torch.bfloat16 and fp8_enabled() and fp8.step_active:
            a_cur = fp8.site_ptrs(fp8_dy_site, x.device)[1]
            K.check(K.lib.lavt_norm_bwd_apply_amax(K.ptr(dy), K.ptr(x), K.ptr(y), K.ptr(mean), K.ptr(rstd), K.ptr(_f32(gamma)), K.ptr(_f32(beta)),
                                                   None, 1, K.ptr(s[0]), K.ptr(s[1]), count, K.ptr(dx), None, a_cur, 1, R, Cc, K.stream()))
            fp8.put_dy_amax(dx, a_cur)
            return dx
        K.check(K.lib.lavt_norm_bwd_apply(K.dt(x.dtype), K.ptr(dy), K.ptr(x), K.ptr(y), K.ptr(mean), K.ptr(rstd), K.ptr(_f32(gamma)), K.ptr(_f32(beta)),
                                          None, 1, K.ptr(s[0]), K.ptr(s[1]), count, K.ptr(dx), None, 1, R, Cc, K.stream()))
        return dx


@K.scoped
class _BatchNormRelu(torch.autograd.Function):
    """BatchNorm2d + ReLU on NHWC rows [R, C].  training: batch statistics (exchanged over `group` when given =
    SyncBatchNorm semantics, train.py:589), running stats updated in place; eval: running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, group, kern, fp8_site=None, fp8_dy_site=None):
        x = x.contiguous()
        R, Cc = x.shape
        count = float(R)
        if training and group is None and hasattr(kern, "stats_fused"):
            mean, rstd = kern.stats_fused(x, eps, running_mean, running_var, momentum)
        elif training:
            s = kern.stats(x)                                # [sum x, centred second moment] of the local rows
            if group is not None and hasattr(kern, "combine_finalize"):
                # gather, then ONE kernel: parallel-variance combination of the ranks' pairs + mean / rstd / running estimates
                allst = syncbn_gather(s, group)
                count = float(R * allst.shape[0])
                mean, rstd = kern.combine_finalize(allst, R, eps, running_mean, running_var, momentum)
            else:
                if group is not None:
                    s, count = syncbn_exchange_forward(s, R, group)
                mean, rstd = kern.finalize(s.view(2, Cc), count, eps, running_mean, running_var, momentum)
        else:
            # eval: mean = running_mean, var = running_var  (sum = mean, m2 = var, count = 1)
            mean, rstd = kern.finalize(torch.stack([running_mean, running_var]), 1.0, eps, None, None, 0.0)
        y = kern.apply(x, mean, rstd, gamma, beta, fp8_site) if fp8_site is not None else kern.apply(x, mean, rstd, gamma, beta)
        ctx.save_for_backward(x, y, gamma, beta, mean, rstd)
        ctx.cfg = (training, count, group, kern)
        ctx.fp8_dy_site = fp8_dy_site
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, rstd = ctx.saved_tensors
        training, count, group, kern = ctx.cfg
        dy = dy.contiguous()
        if training and getattr(kern, "fused_sinks", False):
            # the two LOCAL sums are d beta / d gamma -- accumulate them straight into the parameters' gradient sinks (when the step harness
            # provides them): no clones, no AccumulateGrad adds.  Single rank: the apply pass reads them from there; SyncBatchNorm: the apply
            # pass needs the sums over all ranks -- a stacked copy goes through ONE all-reduce (the parameters keep the local sums: DDP averages
            # parameter gradients later)
            bbuf, bsink = sinks.buf(beta, (x.shape[1],))
            gbuf, gsink = sinks.buf(gamma, (x.shape[1],))
            s = kern.bwd_stats(dy, x, y, mean, rstd, gamma, beta, out=(bbuf, gbuf))
            if group is not None:
                s = syncbn_exchange_backward(torch.stack([bbuf, gbuf]), group)
            dx = (kern.bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count, ctx.fp8_dy_site) if ctx.fp8_dy_site is not None
                  else kern.bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count))
            return dx, sinks.done(gamma, gbuf, gsink), sinks.done(beta, bbuf, bsink), None, None, None, None, None, None, None, None, None
        s0, s1 = kern.bwd_stats(dy, x, y, mean, rstd, gamma, beta)
        s = torch.stack([s0, s1])
        dgamma, dbeta = s[1].clone(), s[0].clone()          # local sums: DDP averages parameter grads later
        if not training:
            s.zero_()                                       # running statistics are constants: no batch terms
        elif group is not None:
            syncbn_exchange_backward(s, group)
        dx = (kern.bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count, ctx.fp8_dy_site) if ctx.fp8_dy_site is not None
              else kern.bwd_apply(dy, x, y, mean, rstd, gamma, beta, s, count))
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def batch_norm_relu(x, bn: torch.nn.modules.batchnorm._BatchNorm, track=True, fp8_site=None, fp8_dy_site=None):
    """`bn` is the module holding the parameters/buffers (nn.BatchNorm2d, or nn.SyncBatchNorm after
    convert_sync_batchnorm -> statistics are all-reduced over its process group).
    track=False: the caller counts the batch itself (bn_count_batches: one launch for all its layers instead of one `add_` each).
    fp8_site / fp8_dy_site (fp8_act_site / fp8_dy_site of the consuming / producing convolution; configs[4]): the apply pass writes the e4m3 twin of the
    output, the backward apply pass records |max| of the input gradient -- the quantiser / |max| launches of those convolutions disappear."""
    import torch.distributed as dist
    group = None
    if isinstance(bn, torch.nn.SyncBatchNorm) and bn.training and dist.is_available() and dist.is_initialized():
        group = bn.process_group if bn.process_group is not None else dist.group.WORLD
        from .ddp import FORCE_COLLECTIVES
        if dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES:
            group = None
    training = bn.training or bn.running_mean is None
    if track and training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BatchNormRelu.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                                bn.momentum if bn.momentum is not None else 0.1, bn.eps, group, _HipBnKernels, fp8_site, fp8_dy_site)


def bn_count_batches(bns):
    """num_batches_tracked += 1 of every training-mode layer in `bns`, as ONE multi-tensor launch (six single-workgroup `add_` launches of 4.7 us each
    sat on the decoder's forward chain)"""
    ts = [bn.num_batches_tracked for bn in bns if (bn.training or bn.running_mean is None) and bn.num_batches_tracked is not None]
    if ts:
        torch._foreach_add_(ts, 1)


# ------------------------------------------------------------------------------------------ language gate
@K.scoped
class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gpre, r):
        x, gpre, r = x.contiguous(), gpre.contiguous(), r.contiguous()
        out = torch.empty_like(x)
        K.check(K.lib.lavt_gate_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(gpre), K.ptr(r), K.ptr(out), x.numel(), K.stream()))
        ctx.save_for_backward(gpre, r)
        return out

    @staticmethod
    def backward(ctx, d):
        gpre, r = ctx.saved_tensors
        d = d.contiguous()
        dg, dr = torch.empty_like(gpre), torch.empty_like(r)
        K.check(K.lib.lavt_gate_bwd(K.dt(d.dtype), K.ptr(d), K.ptr(gpre), K.ptr(r), None, K.ptr(dg), K.ptr(dr), d.numel(), K.stream()))
        return d, dg, dr


def gate(x, gpre, r):
    """x + tanh(gpre) * r   (lib/backbone.py:669 with the Tanh of res_gate folded in)"""
    return _Gate.apply(x, gpre, r)


# ------------------------------------------------------------------------------------------ PWAM pixel-word attention
@K.scoped
class _PwamAttn(torch.autograd.Function):
    """softmax_words(q k^T * C^-1/2 + maskbias) v   per sample and head group (lib/backbone.py:1349-1363).
    q [B*T, C]; k, v [B*KV_LD, C] (rows >= n_l are zero; may be column blocks of a wider matrix: row stride kv_ld); maskbias fp32 [B, KV_LD].
    `sinks_kv` = (dk view, dv view) of a caller-owned gradient buffer: the key / value gradients are written there (the hoisted
    all-stages K / V projection reads them as ONE matrix) and the same views are returned as the gradients."""

    @staticmethod
    def forward(ctx, q, k, v, maskbias, B, T, n_l, G, sinks_kv):
        q = q.contiguous()
        assert k.stride(1) == 1 and v.stride(1) == 1 and k.stride(0) == v.stride(0)
        dtype, dev = q.dtype, q.device
        Cc = q.shape[1]
        kld = k.stride(0)
        c = Cc // G
        alpha = float(Cc ** -0.5)
        o = torch.empty_like(q)
        Ps = []
        for g in range(G):
            S = torch.empty(B * T, KV_LD, dtype=dtype, device=dev)
            gemm_nt(dtype, T, KV_LD, c, q, Cc, k, kld, S, KV_LD, batch=B, strideA=T * Cc, strideB=KV_LD * kld, strideC=T * KV_LD,
                    alpha=alpha, bias=maskbias, strideBias=KV_LD, a_off=g * c, b_off=g * c)
            P = torch.empty_like(S)
            K.check(K.lib.lavt_rowsoftmax_fwd(K.dt(dtype), K.ptr(S), K.ptr(P), B * T, n_l, KV_LD, K.stream()))
            gemm_nt(dtype, T, c, KV_LD, P, KV_LD, v, kld, o, Cc, batch=B, strideA=T * KV_LD, strideB=KV_LD * kld, strideC=T * Cc,
                    b_kmajor=True, b_off=g * c, c_off=g * c)
            Ps.append(P)
        ctx.save_for_backward(q, k, v, *Ps)
        ctx.dims = (B, T, n_l, G, Cc, alpha, kld)
        ctx.sinks_kv = sinks_kv
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, *Ps = ctx.saved_tensors
        B, T, n_l, G, Cc, alpha, kld = ctx.dims
        dtype, dev = q.dtype, q.device
        c = Cc // G
        do = do.contiguous()
        dq = torch.empty_like(q)
        dk = torch.zeros(B * KV_LD, Cc, dtype=torch.float32, device=dev)
        dv = torch.zeros_like(dk)
        for g in range(G):
            P = Ps[g]
            dP = torch.empty_like(P)
            gemm_nt(dtype, T, KV_LD, c, do, Cc, v, kld, dP, KV_LD, batch=B, strideA=T * Cc, strideB=KV_LD * kld, strideC=T * KV_LD,
                    a_off=g * c, b_off=g * c)
            dS = torch.empty_like(P)
            K.check(K.lib.lavt_rowsoftmax_bwd(K.dt(dtype), K.ptr(P), K.ptr(dP), K.ptr(dS), B * T, n_l, KV_LD, K.stream()))
            gemm_nt(dtype, T, c, KV_LD, dS, KV_LD, k, kld, dq, Cc, batch=B, strideA=T * KV_LD, strideB=KV_LD * kld, strideC=T * Cc,
                    b_kmajor=True, alpha=alpha, b_off=g * c, c_off=g * c)
            gemm_tn(dtype, KV_LD, c, T, dS, KV_LD, q, Cc, dk, Cc, batch=B, strideA=T * KV_LD, strideB=T * Cc, strideC=KV_LD * Cc,
                    alpha=alpha, b_off=g * c, c_off=g * c)
            gemm_tn(dtype, KV_LD, c, T, P, KV_LD, do, Cc, dv, Cc, batch=B, strideA=T * KV_LD, strideB=T * Cc, strideC=KV_LD * Cc,
                    b_off=g * c, c_off=g * c)
        if ctx.sinks_kv is not None:          # fp32 -> compute dtype straight into the column blocks of the shared key / value gradient matrix
            sk, sv = ctx.sinks_kv
            sk.copy_(dk)
            sv.copy_(dv)
            return dq, sk, sv, None, None, None, None, None, None
        return dq, cast(dk, dtype), cast(dv, dtype), None, None, None, None, None, None


def pwam_attention(q, k, v, maskbias, B, T, n_l, G, sinks_kv=None):
    return _PwamAttn.apply(q, k, v, maskbias, B, T, n_l, G, sinks_kv)


@K.scoped
class _PwamGate(torch.autograd.Function):
    """PWAM + language gate as ONE autograd node on the fused bf16 kernels (csrc/pwam.hip; reference lib/backbone.py:1265-1278, 1329-1372, 604-611, 669):

        vis = GELU(x Wv^T + bv);  P = softmax_words(IN_T(x Wq^T + bq) K^T C^-1/2 + mask);  what = IN_T((P V) Wo^T + bo) = (P - Pbar) VW'
        r = GELU((vis * what) Wm^T + bm);  xg = x + tanh(ReLU(r W1^T) W2^T) * r                                   -> (r, xg)

    Forward, 10 launches: [vis | q] GEMM over the stacked weight, q statistics (2), word kernel (P), word second moments (TN GEMM), language
    kernel (VW', beta), mix kernel (mm), project_mm GEMM, two gate GEMMs (the second writes xg in its epilogue).  bq and bo sit in front of an
    instance norm: they change nothing and their gradients are exactly zero.  Backward: gate_bwd, three data-gradient GEMMs (ReLU' / GELU' in
    the epilogues), mix A (d vpre, d what), H = dwhat^T P (TN), language kernel 1, word kernel (dS), G = dS^T q (TN), language kernel 2, mix C
    (dq), one data-gradient GEMM over [d vpre | dq], dV GEMM + the weight gradients (grouped)."""

    @staticmethod
    def forward(ctx, x, k, v, maskbias, kv_sinks, dims, Wv, bv, Wq, bq, Wo, bo, Wm, bm, W1, W2):
        B, T, n_l = dims
        x = x.contiguous()
        dtype, dev = x.dtype, x.device
        assert dtype == torch.bfloat16
        M, Cc = x.shape
        assert k.stride(1) == 1 and v.stride(1) == 1 and k.stride(0) == v.stride(0)
        kld = k.stride(0)
        alpha = float(Cc ** -0.5)
        Wst = weights.get_cat((Wv, Wq), dtype)                      # [2C, C]
        vpre = torch.empty(M, Cc, dtype=dtype, device=dev)          # x Wv^T (the bias joins in the mix kernel)
        q = torch.empty_like(vpre)
        gemm_nt(dtype, M, 2 * Cc, Cc, x, Cc, Wst, Cc, vpre, Cc, C2=q, ldc2=Cc, c_split=Cc)
        mean = torch.empty(B, Cc, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        ws = _scratch(1025 * B * 2 * Cc, dev)
        _note(f"in-stats {M}x{Cc}")
        K.check(K.lib.lavt_colstats_meanrstd(K.dt(dtype), K.ptr(q), K.ptr(mean), K.ptr(rstd), K.ptr(ws), ws.numel(), B, T, Cc, 1e-5, None, None, 0.0, K.stream()))
        P = torch.empty(M, KV_LD, dtype=dtype, device=dev)
        _note(f"words {M}x{Cc}", 2.0 * M * Cc * KV_LD)
        lf = torch.empty(B * (2 * Cc + KV_LD + KV_LD * KV_LD), dtype=torch.float32, device=dev)
        beta, rw, pbar, cov = lf[:B * Cc], lf[B * Cc:2 * B * Cc], lf[2 * B * Cc:2 * B * Cc + B * KV_LD], lf[2 * B * Cc + B * KV_LD:]
        rec, nrec = None, 0
        if _PWAM_RECORDS:
            # P^T P, colsum(P) as a by-product of the word kernel: per-workgroup records that the language kernel adds (no P^T P launch, no reduction launch)
            nrec = int(K.lib.lavt_pwam_words_records(B, T, Cc))
            rec = _scratch(B * nrec * (KV_LD * KV_LD + KV_LD), dev)
            K.check(K.lib.lavt_pwam_words_fwd_moments(K.ptr(q), Cc, K.ptr(k), kld, K.ptr(mean), K.ptr(rstd), K.ptr(maskbias), K.ptr(P), K.ptr(rec), B, T, Cc, n_l, alpha, K.stream()))
            PP = sumP = None
        else:
            K.check(K.lib.lavt_pwam_words_fwd(K.ptr(q), Cc, K.ptr(k), kld, K.ptr(mean), K.ptr(rstd), K.ptr(maskbias), K.ptr(P), B, T, Cc, n_l, alpha, K.stream()))
            st = zero_arena.take(B * (KV_LD * KV_LD + KV_LD), torch.float32, dev)
            PP, sumP = st[:B * KV_LD * KV_LD], st[B * KV_LD * KV_LD:]
            gemm_tn(dtype, KV_LD, KV_LD, T, P, KV_LD, P, KV_LD, PP, KV_LD, batch=B, strideA=T * KV_LD, strideB=T * KV_LD, strideC=KV_LD * KV_LD, colsum=sumP, strideColsum=KV_LD)
        VWc = torch.empty(B, Cc, KV_LD, dtype=dtype, device=dev)
        VWw = torch.empty(B, KV_LD, Cc, dtype=dtype, device=dev)
        mm = torch.empty_like(vpre)
        _note(f"lang {B}x{Cc}")
        K.check(K.lib.lavt_pwam_lang_fwd_records(K.ptr(v), kld, K.ptr(weights.get(Wo, dtype, "lin")), K.ptr(PP), K.ptr(sumP), K.ptr(rec), nrec, K.ptr(VWc), K.ptr(VWw),
                                                 K.ptr(beta), K.ptr(rw), K.ptr(pbar), K.ptr(cov), B, T, Cc, 1e-5, K.stream()))
        _note(f"mix0 {M}x{Cc}", 2.0 * M * Cc * KV_LD)
        K.check(K.lib.lavt_pwam_mix(0, K.ptr(P), K.ptr(VWc), K.ptr(beta), None, K.ptr(_f32(bv)), K.ptr(vpre), Cc, None, 0, K.ptr(mm), Cc, None, 0, B, T, Cc, K.stream()))
        rpre = torch.empty_like(vpre)
        r = torch.empty_like(vpre)
        gemm_nt(dtype, M, Cc, Cc, mm, Cc, weights.get(Wm, dtype, "lin"), Cc, r, Cc, bias=_f32(bm), act=K.ACT_GELU, Cpre=rpre, ldcpre=Cc)
        g1 = torch.empty_like(vpre)
        gemm_nt(dtype, M, Cc, Cc, r, Cc, weights.get(W1, dtype, "lin"), Cc, g1, Cc, act=K.ACT_RELU)
        g2 = torch.empty_like(vpre)
        xg = torch.empty_like(vpre)
        gemm_nt(dtype, M, Cc, Cc, g1, Cc, weights.get(W2, dtype, "lin"), Cc, xg, Cc, act=K.ACT_TANH, Cpre=g2, ldcpre=Cc, R=x, ldr=Cc, mul=r, ldmul=Cc)
        ctx.save_for_backward(x, k, v, vpre, q, mean, rstd, P, VWc, VWw, beta, rw, pbar, cov, mm, rpre, r, g1, g2, Wv, bv, Wq, bq, Wo, bo, Wm, bm, W1, W2)
        ctx.dims, ctx.kv_sinks, ctx.alpha = dims, kv_sinks, alpha
        ctx.set_materialize_grads(False)          # an unused output arrives as None in backward (the last stage's gated x)
        return r, xg

    @staticmethod
    def backward(ctx, dr_out, dxg):
        (x, k, v, vpre, q, mean, rstd, P, VWc, VWw, beta, rw, pbar, cov, mm, rpre, r, g1, g2, Wv, bv, Wq, bq, Wo, bo, Wm, bm, W1, W2) = ctx.saved_tensors
        B, T, n_l = ctx.dims
        alpha = ctx.alpha
        dtype, dev = x.dtype, x.device
        M, Cc = x.shape
        kld = k.stride(0)
        gate_live = dxg is not None              # the last stage's gated x feeds nothing (reference lib/backbone.py:669-686): its gate gets no gradient
        dxg = dxg.contiguous() if gate_live else None
        dr_out = dr_out.contiguous() if dr_out is not None else None
        W2c, W1c, Wmc, Woc = (weights.get(w, dtype, "lin") for w in (W2, W1, Wm, Wo))
        Wst = weights.get_cat((Wv, Wq), dtype)
        drpre = torch.empty_like(x)
        if gate_live:
            dg2 = torch.empty_like(x)
            dr = torch.empty_like(x)
            K.check(K.lib.lavt_gate_bwd(K.dt(dtype), K.ptr(dxg), K.ptr(g2), K.ptr(r), K.ptr(dr_out), K.ptr(dg2), K.ptr(dr), x.numel(), K.stream()))
            dpre1 = torch.empty_like(x)
            if Cc % 64 == 0:                      # activation gradients in the data-gradient GEMMs' epilogues
                gemm_nt(dtype, M, Cc, Cc, dg2, Cc, W2c, Cc, dpre1, Cc, b_kmajor=True, dact_pre=g1, lddact=Cc, dact=K.ACT_RELU)
                gemm_nt(dtype, M, Cc, Cc, dpre1, Cc, W1c, Cc, drpre, Cc, b_kmajor=True, dact_pre=rpre, lddact=Cc, dact=K.ACT_GELU, R=dr, ldr=Cc, res_first=True)
            else:                                 # (Swin-T's 96-channel stage: the fused-epilogue kernel needs K % 64 == 0)
                tmp = torch.empty_like(x)
                gemm_nt(dtype, M, Cc, Cc, dg2, Cc, W2c, Cc, tmp, Cc, b_kmajor=True)
                K.check(K.lib.lavt_act_bwd(K.dt(dtype), K.ACT_RELU, K.ptr(tmp), K.ptr(g1), K.ptr(dpre1), x.numel(), K.stream()))
                gemm_nt(dtype, M, Cc, Cc, dpre1, Cc, W1c, Cc, tmp, Cc, b_kmajor=True, R=dr, ldr=Cc)
                K.check(K.lib.lavt_act_bwd(K.dt(dtype), K.ACT_GELU, K.ptr(tmp), K.ptr(rpre), K.ptr(drpre), x.numel(), K.stream()))
        else:
            if dr_out is None:
                dr_out = torch.zeros_like(x)
            K.check(K.lib.lavt_act_bwd(K.dt(dtype), K.ACT_GELU, K.ptr(dr_out), K.ptr(rpre), K.ptr(drpre), x.numel(), K.stream()))
        dmm = torch.empty_like(x)
        gemm_nt(dtype, M, Cc, Cc, drpre, Cc, Wmc, Cc, dmm, Cc, b_kmajor=True)
        g = torch.empty(M, 2 * Cc, dtype=dtype, device=dev)                 # [d vpre | dq]: the A operand of the stacked data / weight gradient
        dwh = torch.empty_like(x)
        _note(f"mix1 {M}x{Cc}", 2.0 * M * Cc * KV_LD)
        Qp = torch.empty(B * int(K.lib.lavt_pwam_q_parts(Cc)) * (KV_LD * KV_LD + KV_LD), dtype=torch.float32, device=dev)      # partial records, written plainly
        dVW = torch.empty(B * KV_LD, Cc, dtype=dtype, device=dev)
        # one zeroed side buffer for the targets of the split reductions (partial tiles, then += the fixed-order sum): G, colsum(dS) (+ H^T, s on the round-5 path)
        if _PWAM_RECORDS:
            # H^T = dwhat^T P and colsum(dwhat) as a by-product of the mix kernel: per-workgroup records that the language kernel adds
            nrec1 = int(K.lib.lavt_pwam_mix1_records(B, T, Cc))
            rec1 = _scratch(B * nrec1 * Cc * (KV_LD + 1), dev)
            K.check(K.lib.lavt_pwam_mix1(K.ptr(P), K.ptr(VWc), K.ptr(beta), K.ptr(_f32(bv)), K.ptr(vpre), Cc, K.ptr(dmm), Cc, K.ptr(g), 2 * Cc, K.ptr(dwh), Cc, K.ptr(rec1),
                                         B, T, Cc, K.stream()))
            z = zero_arena.take(B * (KV_LD * Cc + KV_LD), torch.float32, dev)
            G, sdS = z[:B * KV_LD * Cc], z[B * KV_LD * Cc:]
            _note(f"lang {B}x{Cc}")
            K.check(K.lib.lavt_pwam_lang_bwd1_records(None, None, K.ptr(rec1), nrec1, K.ptr(VWc), K.ptr(rw), K.ptr(pbar), K.ptr(cov), K.ptr(dVW), K.ptr(Qp), B, T, Cc, K.stream()))
        else:
            K.check(K.lib.lavt_pwam_mix(1, K.ptr(P), K.ptr(VWc), K.ptr(beta), None, K.ptr(_f32(bv)), K.ptr(vpre), Cc, K.ptr(dmm), Cc, K.ptr(g), 2 * Cc, K.ptr(dwh), Cc,
                                        B, T, Cc, K.stream()))
            nz = B * (Cc * KV_LD + Cc + KV_LD * Cc + KV_LD)
            z = zero_arena.take(nz, torch.float32, dev)
            o = 0
            HT = z[o:o + B * Cc * KV_LD]; o += B * Cc * KV_LD
            s = z[o:o + B * Cc]; o += B * Cc
            G = z[o:o + B * KV_LD * Cc]; o += B * KV_LD * Cc
            sdS = z[o:o + B * KV_LD]
            gemm_tn(dtype, Cc, KV_LD, T, dwh, Cc, P, KV_LD, HT, KV_LD, batch=B, strideA=T * Cc, strideB=T * KV_LD, strideC=Cc * KV_LD, colsum=s, strideColsum=Cc)
            _note(f"lang {B}x{Cc}")
            K.check(K.lib.lavt_pwam_lang_bwd1(K.ptr(HT), K.ptr(s), K.ptr(VWc), K.ptr(rw), K.ptr(pbar), K.ptr(cov), K.ptr(dVW), K.ptr(Qp), B, T, Cc, K.stream()))
        dS = torch.empty_like(P)
        _note(f"words {M}x{Cc}", 2.0 * M * (Cc + KV_LD) * KV_LD)
        K.check(K.lib.lavt_pwam_words_bwd(K.ptr(dwh), Cc, K.ptr(VWw), K.ptr(Qp), K.ptr(pbar), K.ptr(P), K.ptr(dS), B, T, Cc, K.stream()))
        gemm_tn(dtype, KV_LD, Cc, T, dS, KV_LD, q, Cc, G, Cc, batch=B, strideA=T * KV_LD, strideB=T * Cc, strideC=KV_LD * Cc, colsum=sdS, strideColsum=KV_LD)
        if ctx.kv_sinks is not None:
            dk, dv = ctx.kv_sinks                                            # column blocks of the shared key / value gradient matrix
        else:
            dk = torch.empty(B * KV_LD, Cc, dtype=dtype, device=dev)
            dv = torch.empty_like(dk)
        K2c = torch.empty(B, Cc, KV_LD, dtype=dtype, device=dev)
        cc = torch.empty(2, B, Cc, dtype=torch.float32, device=dev)
        _note(f"lang {B}x{Cc}")
        K.check(K.lib.lavt_pwam_lang_bwd2(K.ptr(G), K.ptr(sdS), K.ptr(k), kld, K.ptr(mean), K.ptr(rstd), K.ptr(dk), dk.stride(0), K.ptr(K2c), K.ptr(cc[0]), K.ptr(cc[1]),
                                          B, T, Cc, alpha, K.stream()))
        _note(f"mix2 {M}x{Cc}", 2.0 * M * Cc * KV_LD)
        K.check(K.lib.lavt_pwam_mix(2, K.ptr(dS), K.ptr(K2c), K.ptr(cc[0]), K.ptr(cc[1]), None, K.ptr(q), Cc, None, 0, g.data_ptr() + 2 * Cc, 2 * Cc, None, 0,
                                    B, T, Cc, K.stream()))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_nt(dtype, M, Cc, 2 * Cc, g, 2 * Cc, Wst, Cc, dx, Cc, b_kmajor=True, R=dxg, ldr=Cc)
        gemm_nt(dtype, B * KV_LD, Cc, Cc, dVW, Cc, Woc, Cc, dv, dv.stride(0), b_kmajor=True)          # dV = dVW Wo
        # ---- weight gradients (joined into grouped launches under the step harness) ----
        grads = {}

        def wgrad(w, b, A, lda, a_off, Bm, ldb, n, kd, rows):
            wbuf, wsink = sinks.buf(w, (n, kd))
            bbuf, bsink = sinks.buf(b, (n,)) if b is not None else (None, True)
            if wgrads.active() and wsink and bsink:
                gemm_tn(dtype, n, kd, rows, A, lda, Bm, ldb, wbuf, kd, colsum=bbuf, a_off=a_off, defer=wgrads)
                wgrads.notify(w)
                if b is not None:
                    wgrads.notify(b)
                grads[id(w)] = None
                if b is not None:
                    grads[id(b)] = None
            else:
                gemm_tn(dtype, n, kd, rows, A, lda, Bm, ldb, wbuf, kd, colsum=bbuf, a_off=a_off)
                grads[id(w)] = sinks.done(w, wbuf, wsink)
                if b is not None:
                    grads[id(b)] = sinks.done(b, bbuf, bsink)

        grads[id(W1)] = grads[id(W2)] = None
        if gate_live:
            wgrad(W2, None, dg2, Cc, 0, g1, Cc, Cc, Cc, M)
            wgrad(W1, None, dpre1, Cc, 0, r, Cc, Cc, Cc, M)
        wgrad(Wm, bm, drpre, Cc, 0, mm, Cc, Cc, Cc, M)
        wgrad(Wv, bv, g, 2 * Cc, 0, x, Cc, Cc, Cc, M)
        wgrad(Wq, None, g, 2 * Cc, Cc, x, Cc, Cc, Cc, M)
        wgrad(Wo, None, dVW, Cc, 0, v, kld, Cc, Cc, B * KV_LD)
        for b0 in (bq, bo):                                                 # in front of an instance norm: exactly zero
            zb, zs = sinks.buf(b0, tuple(b0.shape))
            grads[id(b0)] = sinks.done(b0, zb, zs)
        gp = [grads[id(p)] for p in (Wv, bv, Wq, bq, Wo, bo, Wm, bm, W1, W2)]
        return (dx, dk, dv, None, None, None, *gp)


def pwam_gate(x, k, v, maskbias, kv_sinks, B, T, n_l, fusion, res_gate):
    """fusion: the PWAM module (parameter container), res_gate: nn.Sequential(Linear, ReLU, Linear, Tanh) -> (r, x + tanh(gate(r)) * r)"""
    sila = fusion.image_lang_att
    return _PwamGate.apply(x, k, v, maskbias, kv_sinks, (B, T, n_l), fusion.vis_project[0].weight, fusion.vis_project[0].bias, sila.f_query[0].weight,
                           sila.f_query[0].bias, sila.W[0].weight, sila.W[0].bias, fusion.project_mm[0].weight, fusion.project_mm[0].bias,
                           res_gate[0].weight, res_gate[2].weight)


def pwam_fused_ok(x, G):
    """the fused node covers the default configuration: bf16, one fusion head, channels a multiple of 32"""
    return x.dtype == torch.bfloat16 and G == 1 and x.shape[1] % 32 == 0 and os.environ.get("LAVT_PWAM_FUSED", "1") != "0"


@K.scoped
class _KvAll(torch.autograd.Function):
    """The language key / value projections of ALL stages (f_key / f_value of the four PWAMs: Conv1d(768 -> C_i, 1) on the same <= 22 word
    features, masked; lib/backbone.py:1345-1348 per stage) as ONE GEMM over the stacked weight [sum 2 C_i, 768], once per forward, and one
    data-gradient GEMM in backward (was 8 + 8 launches plus 7 gradient adds).  Outputs: per layer a column block [B*KV_LD, C_i] of one
    matrix (row stride = total width; rows >= n_l zero).  The consumers (_PwamAttn) write their key / value gradients into the matching
    column blocks of `ctx_kv.grad` and return those views."""

    @staticmethod
    def forward(ctx, lt, ctx_kv, *wb):
        lt = lt.contiguous()
        dtype = lt.dtype
        ws, bs = wb[0::2], wb[1::2]
        Wc = weights.get_cat(ws, dtype)
        Nt, Kd = Wc.shape
        M = lt.shape[0]
        rows = ctx_kv.B * KV_LD
        big = zero_arena.take((rows, Nt), dtype, lt.device)
        bias = weights.get_bias_cat(bs)
        gemm_nt(dtype, M, Nt, Kd, lt, Kd, Wc, Kd, big, Nt, bias=bias, row_scale=ctx_kv.mask_rows, c_rowmap=ctx_kv.kv_map)
        ctx.save_for_backward(lt, *wb)
        ctx.ctx_kv = ctx_kv
        ctx_kv.grad = torch.empty_like(big) if any(ctx.needs_input_grad) else None       # every column block is written by its consumer's backward
        outs, off = [], 0
        for w in ws:
            outs.append(big[:, off:off + w.shape[0]])
            off += w.shape[0]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lt, *wb = ctx.saved_tensors
        ws, bs = wb[0::2], wb[1::2]
        kv = ctx.ctx_kv
        dtype = lt.dtype
        G = kv.grad
        Nt, Kd, M = G.shape[1], lt.shape[1], lt.shape[0]
        off = 0
        for w, g in zip(ws, grads):               # consumers return the views of G they filled; anything else is copied in (or zero)
            n = w.shape[0]
            if g is None:
                G[:, off:off + n].zero_()
            elif g.data_ptr() != G.data_ptr() + off * G.element_size() or g.stride(0) != Nt:
                G[:, off:off + n].copy_(g)
            off += n
        Wc = weights.get_cat(ws, dtype)
        dlt = None
        if ctx.needs_input_grad[0]:
            dlt = torch.empty_like(lt)
            gemm_nt(dtype, M, Kd, Nt, G, Nt, Wc, Kd, dlt, Kd, a_rowmap=kv.kv_map, b_kmajor=True, row_scale=kv.mask_rows)
        out, off = [], 0
        for w, b in zip(ws, bs):
            n = w.shape[0]
            wbuf, wsink = sinks.buf(w, (n, Kd))
            bbuf, bsink = sinks.buf(b, (n,))
            kw = dict(a_rowmap=kv.kv_map, a_rowscale=kv.mask_rows, a_rowscale_binary=True, alpha=1.0, colsum=bbuf, a_off=off)
            if wgrads.active() and wsink and bsink and dtype == torch.bfloat16:
                gemm_tn(dtype, n, Kd, M, G, Nt, lt, Kd, wbuf, Kd, defer=wgrads, **kw)
                wgrads.notify(w)
                wgrads.notify(b)
                out += [None, None]
            else:
                gemm_tn(dtype, n, Kd, M, G, Nt, lt, Kd, wbuf, Kd, **kw)
                out += [sinks.done(w, wbuf, wsink), sinks.done(b, bbuf, bsink)]
            off += n
        return (dlt, None, *out)


def kv_all(lt, ctx_kv, layers):
    """layers: [(f_key conv, f_value conv), ...] -> list of (k, v, (dk sink, dv sink) or None) per entry"""
    wb = []
    for fk, fv in layers:
        wb += [fk.weight, fk.bias, fv.weight, fv.bias]
    # per-call holder: the gradient matrix belongs to THIS forward (a second forward before the backward must not replace it)
    ctx_kv = types.SimpleNamespace(B=ctx_kv.B, mask_rows=ctx_kv.mask_rows, kv_map=ctx_kv.kv_map, grad=None)
    outs = _KvAll.apply(lt, ctx_kv, *wb)
    res, off = [], 0
    for i, (fk, fv) in enumerate(layers):
        k, v = outs[2 * i], outs[2 * i + 1]
        nk, nv = fk.weight.shape[0], fv.weight.shape[0]
        sink = (ctx_kv.grad[:, off:off + nk], ctx_kv.grad[:, off + nk:off + nk + nv]) if ctx_kv.grad is not None else None
        res.append((k, v, sink))
        off += nk + nv
    return res


# ------------------------------------------------------------------------------------------ text side (BERT encoder of lavt_one / lavt_video)
@K.scoped
class _BertEmbed(torch.autograd.Function):
    """word[ids] + token_type[tt] + position[0..N-1] (BertEmbeddings.forward before its LayerNorm; HF transformers 3.0.2 modeling_bert.py)."""

    @staticmethod
    def forward(ctx, ids, tt, word, pos, typ, N, dtype):
        ids = ids.contiguous().view(-1).long()
        tt = tt.contiguous().view(-1).long() if tt is not None else None
        rows, H = ids.numel(), word.shape[1]
        assert N <= pos.shape[0], f"bert_embed: {N} tokens but only {pos.shape[0]} positions"
        out = torch.empty(rows, H, dtype=dtype, device=word.device)
        K.check(K.lib.lavt_bert_embed_fwd(K.dt(dtype), K.ptr(ids), K.ptr(tt), K.ptr(_f32(word)), K.ptr(_f32(pos)), K.ptr(_f32(typ)), K.ptr(out),
                                          rows, N, H, K.stream()))
        ctx.save_for_backward(ids, tt, word, pos, typ)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, dy):
        ids, tt, word, pos, typ = ctx.saved_tensors
        dy = dy.contiguous()
        dw, ws = sinks.buf(word, tuple(word.shape))
        dp, ps = sinks.buf(pos, tuple(pos.shape))
        dt_, ts = sinks.buf(typ, tuple(typ.shape))
        K.check(K.lib.lavt_bert_embed_bwd(K.dt(dy.dtype), K.ptr(dy), K.ptr(ids), K.ptr(tt), K.ptr(dw), K.ptr(dp), K.ptr(dt_), ids.numel(), ctx.N,
                                          word.shape[1], K.stream()))
        return None, None, sinks.done(word, dw, ws), sinks.done(pos, dp, ps), sinks.done(typ, dt_, ts), None, None


def bert_embed(ids, token_type_ids, word, pos, typ, N, dtype):
    return _BertEmbed.apply(ids, token_type_ids, word, pos, typ, N, dtype)


@K.scoped
class _Dropout(torch.autograd.Function):
    """y = dropout(x) (+ residual): the keep mask is drawn with torch's generator (so torch.manual_seed governs it, as for nn.Dropout),
    the scaling / masking / residual add is one HIP kernel."""

    @staticmethod
    def forward(ctx, x, residual, p):
        x = x.contiguous()
        keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - p)
        y = torch.empty_like(x)
        if residual is not None:
            residual = residual.contiguous()
        K.check(K.lib.lavt_dropout(K.dt(x.dtype), K.ptr(x), K.ptr(keep), 1.0 / (1.0 - p), K.ptr(residual), K.ptr(y), x.numel(), K.stream()))
        ctx.save_for_backward(keep)
        ctx.scale, ctx.has_res = 1.0 / (1.0 - p), residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        keep, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        K.check(K.lib.lavt_dropout(K.dt(dy.dtype), K.ptr(dy), K.ptr(keep), ctx.scale, None, K.ptr(dx), dy.numel(), K.stream()))
        return dx, (dy if ctx.has_res else None), None


def dropout(x, p, training, residual=None):
    """nn.Dropout(p)(x) [+ residual]; identity (plus the residual through `linear`'s epilogue upstream) when not training or p == 0"""
    if not training or p <= 0.0:
        assert residual is None, "fold the residual into the producing linear() when dropout is off"
        return x
    return _Dropout.apply(x, residual, float(p))


@K.scoped
class _MaskedSelfAttn(torch.autograd.Function):
    """BertSelfAttention core: softmax(q k^T / sqrt(hd) + keybias) v per (sample, head); qkv [B*N, 3H] token-major (q | k | v), keybias fp32 [B, N]
    (= (1 - attention_mask) * -10000).  Heads are regrouped head-major ([B][heads][Np][q|k|v x hd]) so that each GEMM is ONE batched launch;
    the (sample, head) blocks go through lavt_attn_softmax as one "window" with B*heads "heads", whose dense bias carries the key mask.
    Attention-probability dropout (p > 0 in training) is applied to P before the value GEMM."""

    @staticmethod
    def forward(ctx, qkv, keybias, B, N, heads, p_drop):
        dtype, dev = qkv.dtype, qkv.device
        H = qkv.shape[1] // 3
        hd = H // heads
        Np = -(-N // 8) * 8
        nb = B * heads
        x5 = qkv.view(B, N, 3, heads, hd)
        if Np != N:
            x5 = torch.nn.functional.pad(x5, (0, 0, 0, 0, 0, 0, 0, Np - N))
        qh = x5.permute(0, 3, 1, 2, 4).contiguous().view(nb * Np, 3 * hd)                  # rows (sample, head, token); columns q | k | v
        dense = torch.zeros(B, heads, N, Np, dtype=torch.float32, device=dev)
        dense[..., :N] = keybias.view(B, 1, 1, N)
        scale = float(hd ** -0.5)
        S = torch.empty(nb * Np, Np, dtype=dtype, device=dev)
        gemm_nt(dtype, Np, Np, hd, qh, 3 * hd, qh, 3 * hd, S, Np, batch=nb, strideA=Np * 3 * hd, strideB=Np * 3 * hd, strideC=Np * Np, alpha=scale, b_off=hd)
        P = torch.empty_like(S)
        K.check(K.lib.lavt_attn_softmax_fwd(K.dt(dtype), K.ptr(S), K.ptr(dense), Np, None, 0, K.ptr(P), nb * Np, Np, N, Np, nb, K.stream()))
        keep = None
        Pd = P
        if p_drop > 0.0:
            keep = torch.empty(P.shape, dtype=torch.uint8, device=dev).bernoulli_(1.0 - p_drop)
            Pd = torch.empty_like(P)
            K.check(K.lib.lavt_dropout(K.dt(dtype), K.ptr(P), K.ptr(keep), 1.0 / (1.0 - p_drop), None, K.ptr(Pd), P.numel(), K.stream()))
        oh = torch.empty(nb * Np, hd, dtype=dtype, device=dev)
        gemm_nt(dtype, Np, hd, Np, Pd, Np, qh, 3 * hd, oh, hd, batch=nb, strideA=Np * Np, strideB=Np * 3 * hd, strideC=Np * hd, b_kmajor=True, b_off=2 * hd)
        out = oh.view(B, heads, Np, hd)[:, :, :N].permute(0, 2, 1, 3).reshape(B * N, H)
        ctx.save_for_backward(qh, P, Pd if keep is not None else None, keep)
        ctx.dims = (B, N, Np, heads, hd, scale, p_drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        qh, P, Pd, keep = ctx.saved_tensors
        B, N, Np, heads, hd, scale, p_drop = ctx.dims
        dtype, dev = qh.dtype, qh.device
        nb, H = B * heads, heads * hd
        d4 = dout.reshape(B, N, heads, hd)
        if Np != N:
            d4 = torch.nn.functional.pad(d4, (0, 0, 0, 0, 0, Np - N))
        doh = d4.permute(0, 2, 1, 3).contiguous().view(nb * Np, hd)
        dS = torch.empty_like(P)                 # first dP = dO V^T, then overwritten with dS
        gemm_nt(dtype, Np, Np, hd, doh, hd, qh, 3 * hd, dS, Np, batch=nb, strideA=Np * hd, strideB=Np * 3 * hd, strideC=Np * Np, b_off=2 * hd)
        if keep is not None:
            K.check(K.lib.lavt_dropout(K.dt(dtype), K.ptr(dS), K.ptr(keep), 1.0 / (1.0 - p_drop), None, K.ptr(dS), dS.numel(), K.stream()))
        K.check(K.lib.lavt_attn_softmax_bwd(K.dt(dtype), K.ptr(P), K.ptr(dS), nb * Np, N, Np, K.stream()))
        dqh = torch.empty(nb * Np, 3 * hd, dtype=dtype, device=dev)
        gemm_nt(dtype, Np, hd, Np, dS, Np, qh, 3 * hd, dqh, 3 * hd, batch=nb, strideA=Np * Np, strideB=Np * 3 * hd, strideC=Np * 3 * hd, b_kmajor=True,
                alpha=scale, b_off=hd)
        dkv = torch.zeros(2, nb * Np, hd, dtype=torch.float32, device=dev)
        gemm_tn(dtype, Np, hd, Np, dS, Np, qh, 3 * hd, dkv[0], hd, batch=nb, strideA=Np * Np, strideB=Np * 3 * hd, strideC=Np * hd, alpha=scale)
        gemm_tn(dtype, Np, hd, Np, Pd if keep is not None else P, Np, doh, hd, dkv[1], hd, batch=nb, strideA=Np * Np, strideB=Np * hd, strideC=Np * hd)
        dqh.view(nb * Np, 3, hd)[:, 1:] = dkv.permute(1, 0, 2).to(dtype)
        dqkv = dqh.view(B, heads, Np, 3, hd)[:, :, :N].permute(0, 2, 3, 1, 4).reshape(B * N, 3 * H)
        return dqkv, None, None, None, None, None


def masked_self_attention(qkv, keybias, B, N, heads, p_drop=0.0):
    """qkv [B*N, 3H]: columns q | k | v, each heads x head_dim (the layout of `linear_cat(x, (query, key, value))`)"""
    return _MaskedSelfAttn.apply(qkv, keybias, B, N, heads, float(p_drop))


# ------------------------------------------------------------------------------------------ layout changes
@K.scoped
class _Transpose(torch.autograd.Function):
    """[B, R, Cc] -> [B, Cc, R] (both contiguous) with optional dtype change; used for NCHW<->NHWC at the boundary."""

    @staticmethod
    def forward(ctx, x, out_dtype):
        x = x.contiguous()
        B, R, Cc = x.shape
        y = torch.empty(B, Cc, R, dtype=out_dtype, device=x.device)
        # nchw_to_nhwc(B, C=R, HW=Cc): src [B][R][Cc] -> dst [B][Cc][R]
        K.check(K.lib.lavt_nchw_to_nhwc(K.dt(x.dtype), K.ptr(x), K.dt(out_dtype), K.ptr(y), B, R, Cc, K.stream()))
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, Cc, R = dy.shape
        dx = torch.empty(B, R, Cc, dtype=ctx.in_dtype, device=dy.device)
        K.check(K.lib.lavt_nchw_to_nhwc(K.dt(dy.dtype), K.ptr(dy), K.dt(ctx.in_dtype), K.ptr(dx), B, Cc, R, K.stream()))
        return dx, None


def transpose_last2(x, out_dtype=None):
    return _Transpose.apply(x, out_dtype or x.dtype)


@K.scoped
class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        return cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, dy):
        return cast(dy.contiguous(), ctx.in_dtype), None


def cast_ad(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


# ------------------------------------------------------------------------------------------ patch embed
@K.scoped
class _PatchEmbed(torch.autograd.Function):
    """Conv2d(3->C0, k4, s4) on an NCHW fp32 image, zero padded to a multiple of 4 (lib/backbone.py:315-324)."""

    @staticmethod
    def forward(ctx, img, weight, bias, dtype):
        img = img.contiguous().float()
        B, _, H, W = img.shape
        H4, W4 = (H + 3) // 4, (W + 3) // 4
        cols = torch.empty(B * H4 * W4, 48, dtype=dtype, device=img.device)
        K.check(K.lib.lavt_im2col4(K.dt(dtype), K.ptr(img), K.ptr(cols), B, H, W, K.stream()))
        Wc = weights.get(weight, dtype, "lin")
        C0 = Wc.shape[0]
        y = torch.empty(B * H4 * W4, C0, dtype=dtype, device=img.device)
        gemm_nt(dtype, B * H4 * W4, C0, 48, cols, 48, Wc, 48, y, C0, bias=_f32(bias))
        ctx.save_for_backward(cols, weight, bias)
        ctx.dims = (B, H, W, dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        cols, weight, bias = ctx.saved_tensors
        B, H, W, dtype = ctx.dims
        dy = dy.contiguous()
        M, C0 = dy.shape
        dimg = None
        if ctx.needs_input_grad[0]:
            Wc = weights.get(weight, dtype, "lin")
            dcols = torch.empty_like(cols)
            gemm_nt(dtype, M, 48, C0, dy, C0, Wc, 48, dcols, 48, b_kmajor=True)
            dimg = torch.empty(B, 3, H, W, dtype=torch.float32, device=dy.device)
            K.check(K.lib.lavt_col2im4(K.dt(dtype), K.ptr(dcols), K.ptr(dimg), B, H, W, K.stream()))
        dW, wsink = sinks.buf(weight, (C0, 48))
        db, bsink = sinks.buf(bias, (C0,))
        gemm_tn(dtype, C0, 48, M, dy, C0, cols, 48, dW, 48, colsum=db)
        return dimg, sinks.done(weight, dW, wsink), sinks.done(bias, db, bsink), None


def patch_embed(img, weight, bias, dtype):
    return _PatchEmbed.apply(img, weight, bias, dtype)


# ------------------------------------------------------------------------------------------ decoder pieces
@K.scoped
class _ConvTaps(torch.autograd.Function):
    """'same'-padded convolution over NHWC / NDHWC rows as an implicit GEMM: 3x3 (decoder, no bias) or kd x kh x kw with bias
    (Conv3d of SepTPWAM); the input may be the channel concat of x1 and x2; optional fused GELU (pre-activation saved)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, B, D, H, W, act, stats=False):
        x1 = x1.contiguous()
        dtype = x1.dtype
        C1 = x1.shape[1]
        C2 = 0
        if x2 is not None:
            x2 = x2.contiguous()
            C2 = x2.shape[1]
        Wp = weights.get(weight, dtype, "conv3")
        Cout, Cin = weight.shape[0], weight.shape[1]
        ks = tuple(weight.shape[2:])
        kd, kh, kw = (1,) + ks if len(ks) == 2 else ks
        taps = kd * kh * kw
        assert Cin == C1 + C2, f"conv: weight expects {Cin} input channels, got {C1}+{C2}"
        M = B * D * H * W
        y = torch.empty(M, Cout, dtype=dtype, device=x1.device)
        pre = torch.empty_like(y) if act != K.ACT_NONE else None
        x1q = x2q = None
        ctx.fp8_x_amax = 0
        if dtype == torch.bfloat16 and fp8_enabled() and C1 % 16 == 0 and C2 % 16 == 0 and _fp8_conv_fills(M, Cout):
            # configs[4]: e4m3 activations (both concat sources against ONE scale: they feed one contraction) x e4m3 weights on the fp8 MFMA;
            # the bf16 tensors stay saved for the (bf16) backward
            Wq, w_amax = weights.get_fp8(weight, "conv3")
            x1q, a_ptr = fp8.quantize(x1, id(weight))
            x2q = fp8.quantize(x2, id(weight))[0] if x2 is not None else None
            st = gemm_nt(torch.uint8, M, Cout, taps * Cin, x1q, C1, Wq, taps * Cin, y, Cout, A2=x2q, lda2=C2, a_split=C1,
                         conv=(H, W, Cin, 0, D, kd, kh, kw), bias=_f32(bias), act=act, Cpre=pre, ldcpre=Cout, deq=(a_ptr, w_amax.data_ptr()),
                         want_colstats=_CONV_STATS and stats and bias is None and act == K.ACT_NONE)
            if st is not None:          # (the pipelined e4m3 kernel has the statistics epilogue of the bf16 one)
                conv_stats.put(y, st)
            if _FP8_WGRAD and (kd, kh, kw) == (1, 3, 3) and bias is None and D == 1 and K.lib.lavt_conv3x3_wgrad_f8_ok(B, H, W, Cout, Cin, C1 if x2 is not None else Cin):
                ctx.fp8_x_amax = a_ptr          # the e4m3 copies stay alive for the weight gradient (half the bytes of the bf16 tensors beside them)
            else:
                x1q = x2q = None
        elif _conv_split(dtype, M, Cout, Cin, C1, C2, taps, bias, act)[0]:
            # few pixels, long reduction (decoder level 4: 1 800 rows x K = 13 824 = 60-232 tiles walking 72-216 K tiles each): the reduction is cut
            # at tap boundaries over the batch index into fp32 partial outputs, a second small kernel adds them
            sp, over_ch = _conv_split(dtype, M, Cout, Cin, C1, C2, taps, bias, act)
            parts = torch.empty(sp, M, Cout, dtype=torch.float32, device=x1.device)
            if over_ch:
                gemm_nt(dtype, M, Cout, taps * (Cin // sp), x1, C1, Wp, taps * Cin, parts, Cout, A2=x2, lda2=C2, a_split=C1, conv=(H, W, Cin, 0, D, kd, kh, kw),
                        batch=sp, strideC=M * Cout, c_f32=True, conv_kc_split=Cin // sp)
            else:
                gemm_nt(dtype, M, Cout, (taps // sp) * Cin, x1, C1, Wp, taps * Cin, parts, Cout, A2=x2, lda2=C2, a_split=C1, conv=(H, W, Cin, 0, D, kd, kh, kw),
                        batch=sp, strideB=(taps // sp) * Cin, strideC=M * Cout, c_f32=True, conv_tap_split=taps // sp)
            K.check(K.lib.lavt_splitk_reduce(K.dt(dtype), K.ptr(parts), sp, M, Cout, K.ptr(y), Cout, K.stream()))
        else:
            st = gemm_nt(dtype, M, Cout, taps * Cin, x1, C1, Wp, taps * Cin, y, Cout, A2=x2, lda2=C2, a_split=C1,
                         conv=(H, W, Cin, 0, D, kd, kh, kw), bias=_f32(bias), act=act, Cpre=pre, ldcpre=Cout,
                         want_colstats=_CONV_STATS and stats and bias is None and act == K.ACT_NONE)
            if st is not None:
                conv_stats.put(y, st)
        ctx.save_for_backward(x1, x2, weight, bias, pre, x1q if ctx.fp8_x_amax else None, x2q if ctx.fp8_x_amax else None)
        ctx.dims = (B, D, H, W, C1, C2, Cout, kd, kh, kw, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2, weight, bias, pre, x1q, x2q = ctx.saved_tensors
        B, D, H, W, C1, C2, Cout, kd, kh, kw, act = ctx.dims
        dtype = x1.dtype
        dyq = dy_amax = None
        Cin = C1 + C2
        taps = kd * kh * kw
        M = B * D * H * W
        dy = dy.contiguous()
        if act != K.ACT_NONE:
            g = torch.empty_like(dy)
            K.check(K.lib.lavt_act_bwd(K.dt(dtype), act, K.ptr(dy), K.ptr(pre), K.ptr(g), dy.numel(), K.stream()))
            dy = g
        Wp = weights.get(weight, dtype, "conv3")
        dx1 = dx2 = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            dx1 = torch.empty_like(x1)
            dx2 = torch.empty_like(x2) if x2 is not None else None
            if dtype == torch.bfloat16 and fp8_enabled() and _FP8_DGRAD and Cout % 16 == 0 and C1 % 4 == 0 and C2 % 4 == 0 and _fp8_conv_fills(M, C1):
                # configs[4]: e4m3 dY (current scaling: its own |max|, computed in front of the quantiser) x the transposed e4m3 weight copy [Cin][taps][Cout] on the fp8 MFMA; a concat
                # convolution runs as one launch per source (row blocks of the transposed weight), like the bf16 split below
                WqT, w_amax = weights.get_fp8(weight, "conv3t")
                dyq, a_ptr = fp8.quantize_current(dy, (id(weight), "dy"))
                dy_amax = a_ptr
                for dxo, Cn, roff in ((dx1, C1, 0),) + (((dx2, C2, C1),) if x2 is not None else ()):
                    gemm_nt(torch.uint8, M, Cn, taps * Cout, dyq, Cout, WqT, taps * Cout, dxo, Cn, conv=(H, W, Cout, 1, D, kd, kh, kw),
                            b_off=roff * taps * Cout, deq=(a_ptr, w_amax.data_ptr()))
            elif x2 is not None and C1 % 256 == 0 and C2 % 64 == 0 and dtype == torch.bfloat16 and os.environ.get("LAVT_DGRAD_SPLIT", "1") != "0":
                # concat convolution (conv1_2: 512 + 128 input channels): N = 640 is not a multiple of the 256-wide tile, so the whole data gradient
                # fell back to 128x128 tiles (239 us at 2x120x120).  As two launches over column blocks of the packed weight the 512-channel part
                # runs on the 256x256 tile and the skip part on its own.
                for dxo, Cn, boff in ((dx1, C1, 0), (dx2, C2, C1)):
                    sp, over_ch = _conv_split(dtype, M, Cn, Cout, Cout, 0, taps, None, K.ACT_NONE)
                    if sp:
                        parts = torch.empty(sp, M, Cn, dtype=torch.float32, device=dy.device)
                        if over_ch:
                            gemm_nt(dtype, M, Cn, taps * (Cout // sp), dy, Cout, Wp, taps * Cin, parts, Cn, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True,
                                    b_tap_stride=Cin, b_off=boff, batch=sp, strideC=M * Cn, c_f32=True, conv_kc_split=Cout // sp)
                        else:
                            gemm_nt(dtype, M, Cn, (taps // sp) * Cout, dy, Cout, Wp, taps * Cin, parts, Cn, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True,
                                    b_tap_stride=Cin, b_off=boff, batch=sp, strideC=M * Cn, c_f32=True, conv_tap_split=taps // sp)
                        K.check(K.lib.lavt_splitk_reduce(K.dt(dtype), K.ptr(parts), sp, M, Cn, K.ptr(dxo), Cn, K.stream()))
                    else:
                        gemm_nt(dtype, M, Cn, taps * Cout, dy, Cout, Wp, taps * Cin, dxo, Cn, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True, b_tap_stride=Cin,
                                b_off=boff)
            elif x2 is None and _conv_split(dtype, M, Cin, Cout, Cout, 0, taps, None, K.ACT_NONE)[0]:
                sp, over_ch = _conv_split(dtype, M, Cin, Cout, Cout, 0, taps, None, K.ACT_NONE)
                parts = torch.empty(sp, M, Cin, dtype=torch.float32, device=dy.device)
                if over_ch:
                    gemm_nt(dtype, M, Cin, taps * (Cout // sp), dy, Cout, Wp, taps * Cin, parts, Cin, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True,
                            b_tap_stride=Cin, batch=sp, strideC=M * Cin, c_f32=True, conv_kc_split=Cout // sp)
                else:
                    gemm_nt(dtype, M, Cin, (taps // sp) * Cout, dy, Cout, Wp, taps * Cin, parts, Cin, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True,
                            b_tap_stride=Cin, batch=sp, strideC=M * Cin, c_f32=True, conv_tap_split=taps // sp)
                K.check(K.lib.lavt_splitk_reduce(K.dt(dtype), K.ptr(parts), sp, M, Cin, K.ptr(dx1), Cin, K.stream()))
            else:
                gemm_nt(dtype, M, Cin, taps * Cout, dy, Cout, Wp, taps * Cin, dx1, C1, conv=(H, W, Cout, 1, D, kd, kh, kw), b_kmajor=True,
                        b_tap_stride=Cin, C2=dx2, ldc2=C2, c_split=C1)
        dW, wsink = sinks.buf(weight, (Cout, Cin * taps))
        db = bsink = None
        if bias is not None:
            db, bsink = sinks.buf(bias, (Cout,))
        # the GEMM writes [Cout][taps][Cin] (contiguous split-K atomics), a small kernel adds it into the [Cout][Cin][taps] gradient
        def _wgrad():
            ws = 0
            if x1q is not None:
                # configs[4]: e4m3 dY x e4m3 X on the fp8 MFMA (csrc/conv_wgrad.hip, conv_wgrad3x3_f8_kernel): the operands are the copies the forward convolution
                # and the data gradient contracted -- no quantiser launch of its own unless the data gradient stayed in bf16
                ws = int(K.lib.lavt_conv3x3_wgrad_ws(B, H, W, Cout, Cin, C1 if x2 is not None else Cin))
                q, amax = (dyq, dy_amax) if dyq is not None else fp8.quantize_current(dy, (id(weight), "dy"))
                scr = _tn_parts(ws, dy.device)
                if K.prof.enabled:
                    K.prof.note = {"flops": 2.0 * Cout * taps * Cin * M, "shape": f"conv-wgrad-f8 {Cout}x{taps * Cin}x{M}"}
                K.check(K.lib.lavt_conv3x3_wgrad_f8(K.ptr(q), Cout, amax, K.ptr(x1q), C1, K.ptr(x2q), C2, ctx.fp8_x_amax, C1, B, H, W, Cout, Cin, K.ptr(scr), scr.numel(),
                                                    K.ptr(dW), 0, _zero_page(dy.device), K.stream()))
                if wsink:
                    sinks.mark_assigned_ptr(dW.data_ptr())
                return
            if dtype == torch.bfloat16 and (kd, kh, kw) == (1, 3, 3) and bias is None and D == 1 and os.environ.get("LAVT_CONV_WGRAD_TAPS", "1") != "0":
                ws = int(K.lib.lavt_conv3x3_wgrad_ws(B, H, W, Cout, Cin, C1 if x2 is not None else Cin))
            if ws:
                # nine taps fused (csrc/conv_wgrad.hip): X rows in a rolling LDS window, partial tiles through the lent scratch, the reduction kernel
                # accumulates straight into the [Cout][Cin][3][3] gradient -- no packed buffer, no zero fill, no unpack launch
                scr = _tn_parts(ws, dy.device)          # (single-stream scratch: every launch of the step is on the one launch stream)
                if K.prof.enabled:
                    K.prof.note = {"flops": 2.0 * Cout * taps * Cin * M, "shape": f"conv-wgrad {Cout}x{taps * Cin}x{M}"}
                # dW is either the parameter's slice of the zeroed flat gradient buffer (one weight gradient per parameter per step: sinks.buf refuses a
                # second) or fresh zeros: the reduction overwrites instead of adding
                K.check(K.lib.lavt_conv3x3_wgrad(K.ptr(dy), Cout, K.ptr(x1), C1, K.ptr(x2), C2, C1, B, H, W, Cout, Cin, K.ptr(scr), scr.numel(), K.ptr(dW), 0,
                                                 _zero_page(dy.device), K.stream()))
                if wsink:
                    sinks.mark_assigned_ptr(dW.data_ptr())
                return
            packed = torch.zeros(Cout, taps * Cin, dtype=torch.float32, device=dy.device)
            gemm_tn(dtype, Cout, taps * Cin, M, dy, Cout, x1, C1, packed, taps * Cin, B2=x2, ldb2=C2, b_split=C1,
                    conv=(H, W, Cin, D, kd, kh, kw), colsum=db)
            K.check(K.lib.lavt_unpack_conv_grad(K.ptr(packed), K.ptr(dW), Cout, Cin, taps, K.stream()))
        _wgrad()
        return (dx1, dx2, sinks.done(weight, dW, wsink), sinks.done(bias, db, bsink) if bias is not None else None,
                None, None, None, None, None, None)


# (round 5: 4096 -- decoder level 4 at batch 4 has 3600 rows: 11.37 -> 11.32 ms; 2048 was sized on batch 2; tools/r05_knob_sweep2.sh)
_CONV_SPLIT_MAX_ROWS = int(os.environ.get("LAVT_CONV_SPLIT_ROWS", "4096"))


def _conv_split(dtype, M, N, Kc, C1, C2, taps, bias, act):
    """-> (pieces, over_channels): the number of pieces a convolution's reduction is cut into (0 = not split) and whether they are channel blocks
    (lavt_gemm_nt_t.conv_kc_split) or tap groups (conv_tap_split): bf16 tap-walking problems (channels % 64 == 0) without a fused epilogue, few rows
    (<= 2048: the fp32 partials are M x N x pieces x 4 bytes written and re-read) and a long reduction"""
    # (callers ask only for convolutions that stay bf16: in fp8 mode the e4m3 branches are tried first, and the maps they leave in bf16 -- decoder level 4 -- split like in bf16 mode)
    if dtype != torch.bfloat16 or bias is not None or act != K.ACT_NONE or M > _CONV_SPLIT_MAX_ROWS or taps % 3 or taps > 27:
        return 0, False
    if Kc % 64 or C1 % 64 or C2 % 64 or N % 8 or taps * Kc < 4096:
        return 0, False
    d = _kc_pieces(M, N, Kc) if taps == 9 else 0
    if d:
        return d, True
    return (int(os.environ.get("LAVT_CONV_SPLIT_N", "3")) if taps == 9 else taps // 3), False


# Channel-block pieces of a few-pixel 3x3 convolution's reduction (0 = cut at tap boundaries instead: 3 pieces).  Measured on MI355X (tools/conv_small_probe.py,
# 2 x 30 x 30 pixels): what matters is ONE full round of 128x128-tile workgroups on the 256 CUs (the 4-stage ring takes a CU's LDS) -- 1536 -> 512 forward:
# 180 workgroups (3 tap groups) 52-60 us, 240 (4 channel pieces) 45 us, 480 (8) 59 us; 512 -> 512: 33 / 29 / 42 us; data gradient onto 1024 channels: 360 (3 tap
# groups) 56 us, 240 (2 pieces) 39 us, 480 (4) 52 us.  LAVT_CONV_KC_SPLITS: "auto" (largest divisor of the channel blocks that keeps the launch within one round),
# a number (that many pieces where it divides), 0 (off).
_CONV_KC_SPLITS = os.environ.get("LAVT_CONV_KC_SPLITS", "auto")


def _kc_pieces(M, N, Kc):
    if _CONV_KC_SPLITS == "0" or os.environ.get("LAVT_GEMM_PIPE", "2") == "0" or os.environ.get("LAVT_GEMM_GENERAL") is not None or os.environ.get("LAVT_GEMM_V2", "1") == "0" or Kc % 64:
        return 0
    cb, tiles = Kc // 64, -(-M // 128) * -(-N // 128)
    if _CONV_KC_SPLITS != "auto":
        d = int(_CONV_KC_SPLITS)
        return d if d > 1 and cb % d == 0 else 0
    best = max((d for d in range(2, cb + 1) if cb % d == 0 and tiles * d <= 256), default=0)
    return best if best and tiles * best > min(tiles * 3, 256) * 0.9 else 0          # (not worse filled than the 3 tap groups it replaces)


def conv3x3(x1, x2, weight, B, H, W):
    # under autograd (training) the launch also leaves the column statistics of its output for the BatchNorm that follows (ops.conv_stats)
    return _ConvTaps.apply(x1, x2, weight, None, B, 1, H, W, K.ACT_NONE, torch.is_grad_enabled())


def conv3d(x, weight, bias, B, D, H, W, act=K.ACT_NONE):
    """Conv3d(stride 1, 'same' zero padding, kernel sizes 1 or 3 per axis) on NDHWC rows [B*D*H*W, Cin]."""
    if weight.shape[2:] == (1, 1, 1):
        return linear(x, weight, bias, act=act)
    return _ConvTaps.apply(x, None, weight, bias, B, D, H, W, act)


@K.scoped
class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, Hi, Wi, Ho, Wo, fp8_site=None):
        x = x.contiguous()
        Cc = x.shape[1]
        y = torch.empty(B * Ho * Wo, Cc, dtype=x.dtype, device=x.device)
        ctx.dims = (B, Hi, Wi, Ho, Wo, Cc)
        if fp8_site is not None and x.dtype == torch.bfloat16 and fp8_enabled() and fp8.step_active:
            q = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
            a_prev, a_cur = fp8.site_ptrs(fp8_site, x.device)
            K.check(K.lib.lavt_bilinear_fwd_q8(K.ptr(x), K.ptr(y), K.ptr(q), a_prev, a_cur, B, Hi, Wi, Ho, Wo, Cc, K.stream()))
            fp8.put_twin(y, q, a_prev)
            return y
        K.check(K.lib.lavt_bilinear_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(y), B, Hi, Wi, Ho, Wo, Cc, K.stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Hi, Wi, Ho, Wo, Cc = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(B * Hi * Wi, Cc, dtype=dy.dtype, device=dy.device)
        K.check(K.lib.lavt_bilinear_bwd(K.dt(dy.dtype), K.ptr(dy), K.ptr(dx), B, Hi, Wi, Ho, Wo, Cc, K.stream()))
        return dx, None, None, None, None, None, None


def bilinear(x, B, Hi, Wi, Ho, Wo, fp8_site=None):
    """fp8_site: fp8_act_site of the convolution that consumes the output (configs[4]): the kernel writes the e4m3 twin itself"""
    return _Bilinear.apply(x, B, Hi, Wi, Ho, Wo, fp8_site)


@K.scoped
class _ClsHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        R, Cc = x.shape
        y = torch.empty(R, 2, dtype=x.dtype, device=x.device)
        K.check(K.lib.lavt_cls_head_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(_f32(weight)), K.ptr(_f32(bias)), K.ptr(y), R, Cc, K.stream()))
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        R, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw, wsink = sinks.buf(weight, (2, Cc))
        db, bsink = sinks.buf(bias, (2,))
        if wsink and bsink and ln_deferred.active():
            # under the step harness the per-workgroup sums ride in the end-of-backward reduction of the LayerNorm partial sums (no global atomics)
            nblk = int(K.lib.lavt_cls_head_bwd_blocks(K.dt(x.dtype), R, Cc))
            pw, pb = ln_deferred.alloc(nblk * 2 * Cc, x.device), ln_deferred.alloc(nblk * 2, x.device)
            if pw is not None and pb is not None:
                K.check(K.lib.lavt_cls_head_bwd_partial(K.dt(x.dtype), K.ptr(x), K.ptr(dy), K.ptr(_f32(weight)), K.ptr(dx), K.ptr(pw), K.ptr(pb), R, Cc, K.stream()))
                ln_deferred.add(pw, nblk, Cc, dw[0], dw[1], (weight,))
                ln_deferred.add(pb, nblk, 1, db[0:1], db[1:2], (bias,))
                return dx, None, None
        K.check(K.lib.lavt_cls_head_bwd(K.dt(x.dtype), K.ptr(x), K.ptr(dy), K.ptr(_f32(weight)), K.ptr(dx), K.ptr(dw), K.ptr(db), R, Cc, K.stream()))
        return dx, sinks.done(weight, dw, wsink), sinks.done(bias, db, bsink)


def cls_head(x, weight, bias):
    assert weight.shape[0] == 2
    return _ClsHead.apply(x, weight, bias)


@K.scoped
class _LogitsUp(torch.autograd.Function):
    """NHWC [B*Hi*Wi, 2] -> NCHW fp32 [B, 2, Ho, Wo] bilinear, align_corners=True (lib/_utils.py:21)."""

    @staticmethod
    def forward(ctx, x, B, Hi, Wi, Ho, Wo):
        x = x.contiguous()
        y = torch.empty(B, 2, Ho, Wo, dtype=torch.float32, device=x.device)
        K.check(K.lib.lavt_logits_up_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(y), B, Hi, Wi, Ho, Wo, K.stream()))
        ctx.dims = (B, Hi, Wi, Ho, Wo, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Hi, Wi, Ho, Wo, dtype = ctx.dims
        dy = dy.contiguous().float()
        dx = torch.empty(B * Hi * Wi, 2, dtype=dtype, device=dy.device)
        K.check(K.lib.lavt_logits_up_bwd(K.dt(dtype), K.ptr(dy), K.ptr(dx), B, Hi, Wi, Ho, Wo, K.stream()))
        return dx, None, None, None, None, None


def logits_upsample(x, B, Hi, Wi, Ho, Wo):
    return _LogitsUp.apply(x, B, Hi, Wi, Ho, Wo)


# ------------------------------------------------------------------------------------------ fused upsample + weighted CE (+ I/U counts)
@K.scoped
class _UpsampleCE(torch.autograd.Function):
    """loss = F.cross_entropy(F.interpolate(y, (Ho, Wo), bilinear, align_corners=True), target, weight=(w0, w1)) on the low-resolution
    2-class logits rows x [B*Hi*Wi, 2] (lib/_utils.py:21 + losses.py:7-11) without materialising the upsampled logits.
    Returns (loss, stats) with stats = [loss, sum of weights, I, U] (train.py:64-76 pixel counts of the argmax mask)."""

    @staticmethod
    def forward(ctx, x, target, B, Hi, Wi, Ho, Wo, w0, w1):
        x = x.contiguous()
        target = target.contiguous()
        assert target.dtype == torch.int64 and target.numel() == B * Ho * Wo
        out4 = torch.empty(4, dtype=torch.float32, device=x.device)
        ws = _scratch(4 * 2048, x.device)
        K.check(K.lib.lavt_upsample_ce_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(target), float(w0), float(w1), K.ptr(ws), ws.numel(), K.ptr(out4),
                                           B, Hi, Wi, Ho, Wo, K.stream()))
        ctx.dims = (B, Hi, Wi, Ho, Wo, float(w0), float(w1))
        ctx.set_materialize_grads(False)         # (the statistics receive no gradient: None instead of a zero fill per step)
        if wgrads.active():
            # inside the step harness the loss is consumed at once (engine.TrainStep): loss and statistics are views of the four floats that backward
            # reads -- no copy kernel on the captured chain
            stats = out4[:]
            ctx.save_for_backward(x, target, stats)
            ctx.mark_non_differentiable(stats)
            return out4[0], stats
        # public entry (lib._utils.fused_loss in a caller's own loop): the caller may scale the loss in place (`loss /= accum_steps`, a GradScaler) --
        # backward keeps its own four floats, the caller gets copies
        stats = out4.clone()
        ctx.save_for_backward(x, target, out4)
        ctx.mark_non_differentiable(stats)
        return out4[0].clone(), stats

    @staticmethod
    def backward(ctx, dloss, _dstats):
        x, target, out4 = ctx.saved_tensors
        B, Hi, Wi, Ho, Wo, w0, w1 = ctx.dims
        dx = torch.empty_like(x)
        if dloss is None:                        # (nothing downstream of the loss)
            return None, None, None, None, None, None, None, None, None
        dl = dloss.contiguous().float().reshape(1)
        K.check(K.lib.lavt_upsample_ce_bwd(K.dt(x.dtype), K.ptr(x), K.ptr(target), w0, w1, K.ptr(out4), K.ptr(dl), K.ptr(dx), B, Hi, Wi, Ho, Wo, K.stream()))
        return dx, None, None, None, None, None, None, None, None


def upsample_cross_entropy(x, target, B, Hi, Wi, Ho, Wo, weight=(0.9, 1.1)):
    return _UpsampleCE.apply(x, target, B, Hi, Wi, Ho, Wo, weight[0], weight[1])


@K.scoped
class _UpsampleDice(torch.autograd.Function):
    """MultiClassDiceLoss()(F.interpolate(y, (Ho, Wo), bilinear, align_corners=True), target) on the low-resolution 2-class logits rows
    x [B*Hi*Wi, 2] (reference losses.py:38-77 after lib/_utils.py:21), the upsampled logits never written.
    Returns (loss, stats) with stats = [loss, 0, per sample {I0, I1, sum p0^2, sum p1^2, #t==0, #t==1}]."""

    @staticmethod
    def forward(ctx, x, target, B, Hi, Wi, Ho, Wo):
        x = x.contiguous()
        target = target.contiguous()
        assert target.dtype == torch.int64 and target.numel() == B * Ho * Wo
        stats = torch.empty(2 + 6 * B, dtype=torch.float32, device=x.device)
        ws = _scratch(6 * 256 * B, x.device)
        K.check(K.lib.lavt_upsample_dice_fwd(K.dt(x.dtype), K.ptr(x), K.ptr(target), K.ptr(ws), ws.numel(), K.ptr(stats), B, Hi, Wi, Ho, Wo, K.stream()))
        ctx.save_for_backward(x, target, stats)
        ctx.dims = (B, Hi, Wi, Ho, Wo)
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, dloss, _dstats):
        x, target, stats = ctx.saved_tensors
        B, Hi, Wi, Ho, Wo = ctx.dims
        dx = torch.empty_like(x)
        dl = dloss.contiguous().float().reshape(1)
        K.check(K.lib.lavt_upsample_dice_bwd(K.dt(x.dtype), K.ptr(x), K.ptr(target), K.ptr(stats), K.ptr(dl), K.ptr(dx), B, Hi, Wi, Ho, Wo, K.stream()))
        return dx, None, None, None, None, None, None


def upsample_dice_loss(x, target, B, Hi, Wi, Ho, Wo):
    return _UpsampleDice.apply(x, target, B, Hi, Wi, Ho, Wo)
