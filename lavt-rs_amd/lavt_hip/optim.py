"""Fused AdamW + poly learning-rate schedule for the LAVT step (the caller's optimizer, train.py:615-700).

    params_to_optimize = lavt_param_groups(model)                      # the reference's groups: no weight decay on norm / bias-table
    opt = FusedAdamW(params_to_optimize, lr=5e-5, weight_decay=1e-2, total_steps=len(loader) * epochs, power=0.9)
    ... loss.backward(); opt.step()                                    # ONE multi-tensor launch (+ a 1-thread tick) for all tensors

`torch.optim.Optimizer`-shaped (param_groups, state_dict / load_state_dict in torch.optim.AdamW's layout) so that the reference's
checkpoint code (train.py:749-763) keeps working.  The step counter and the schedule live on the device: the optimizer step can be
captured into the same hipGraph as forward/backward and still follows (1 - it/T)^0.9 on every replay (a LambdaLR on the host would be
frozen at capture time).  amsgrad (off by default in the reference) is not implemented.
"""
import os
from typing import Iterable

import torch

from . import _capi as K


def lavt_param_groups(model, text_encoder_layers: int = 10):
    """The reference's parameter groups (train.py:615-660): backbone tensors whose name contains 'norm', 'absolute_pos_embed' or
    'relative_position_bias_table' get weight_decay 0; the rest of the backbone, the classifier and (if the model carries one) the first
    `text_encoder_layers` BERT encoder layers use the default."""
    no_decay, decay = [], []
    for name, p in model.backbone.named_parameters():
        (no_decay if ("norm" in name or "absolute_pos_embed" in name or "relative_position_bias_table" in name) else decay).append(p)
    groups = [{"params": no_decay, "weight_decay": 0.0}, {"params": decay},
              {"params": [p for p in model.classifier.parameters() if p.requires_grad]}]
    enc = getattr(model, "text_encoder", None)
    if enc is not None and hasattr(enc, "encoder"):
        groups.append({"params": [p for i in range(text_encoder_layers) for p in enc.encoder.layer[i].parameters() if p.requires_grad]})
    return groups


def ops_generation():
    """what a descriptor table depends on besides the gradient pointers it lists: the compute copies' buffers (ops.weights.generation) and the layout
    of the flat gradient buffer (ddp.layout_generation: GradBuckets re-lays it out once, in the second zero())"""
    from . import ops, ddp
    return (ops.weights.generation, ddp.layout_generation[0])


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, total_steps=0, power=0.9, context=None):
        """context: the ops.StepContext whose compute-dtype weight copies this optimizer maintains (None = the process-wide default; pass the one given to
        engine.TrainStep when the model runs in a private context)"""
        from . import ops
        self.context = context if context is not None else ops.default_context()
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not implemented (the reference's default is off)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.total_steps, self.power = float(total_steps), float(power)
        self.fuse_copies = os.environ.get("LAVT_ADAMW_FUSE_COPIES", "1") != "0"
        self._tables = None
        self._probe = []
        self._step = None

    # ---- flat optimizer state + device descriptor tables (built lazily: gradients must exist / be re-pointed first) ----
    def _build(self):
        ps = [(g, p) for g in self.param_groups for p in g["params"] if p.requires_grad]
        assert ps, "FusedAdamW: no parameters"
        dev = ps[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdamW runs on GPU memory only (no CPU fallback)")
        total = sum(p.numel() for _, p in ps)
        if self._step is None:
            self._step = torch.zeros(1, dtype=torch.float32, device=dev)
        if not all("exp_avg" in self.state[p] for _, p in ps):
            flat_m, flat_v = torch.zeros(total, dtype=torch.float32, device=dev), torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            for _, p in ps:
                n = p.numel()
                st = self.state[p]
                st.setdefault("step", self._step)
                st["exp_avg"], st["exp_avg_sq"] = flat_m[off:off + n].view_as(p), flat_v[off:off + n].view_as(p)
                off += n
        from . import ops
        desc, hyper, missing, chunks, fused = [], [], [], [], {}
        ce = int(K.lib.lavt_adamw_chunk_elems())
        for g, p in ps:
            if p.grad is None:
                missing.append(p)
                continue
            assert p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
            st = self.state[p]
            # the parameter's bf16 compute copy in the parameter's own layout (Linear / 1x1 weights: lavt_hip.ops.weights, kind 'lin') is written by
            # the update kernel itself; packed conv weights, e4m3 copies and LayerNorm folds are refreshed after it (ops.weights.refresh_all)
            ck = (id(p), torch.bfloat16, "lin")
            ent = ops.weights.store.get(ck) if self.fuse_copies else None
            copy = 0
            if ent is not None and ent[2]() is p and ent[1].numel() == p.numel() and ent[1].is_contiguous():
                copy = ent[1].data_ptr()
                fused[ck] = copy
            for c in range(-(-p.numel() // ce)):
                chunks.append([len(desc), c])
            desc.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), copy])
            hyper.append([g["lr"], g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"]])
        # (the copies' addresses are baked into the table: ops.weights.generation moves whenever one of them gets a new buffer)
        key = tuple(d[1] for d in desc) + tuple(tuple(h) for h in hyper) + (ops_generation(),)
        # (every described parameter with the gradient address the table holds: step(check_tables=False) compares them all -- integer work on the host,
        # no device synchronisation; three probed addresses missed a single re-created .grad)
        self._probe = [(p, p.grad.data_ptr()) for _, p in ps if p.grad is not None]
        self._tables = (key, torch.tensor(desc, dtype=torch.int64).to(dev), torch.tensor(hyper, dtype=torch.float32).to(dev), len(desc),
                        torch.tensor(chunks, dtype=torch.int32).to(dev), len(chunks), dict(fused))

    def _current_key(self):
        out = []
        hy = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.requires_grad and p.grad is not None:
                    out.append(p.grad.data_ptr())
                    hy.append((g["lr"], g["weight_decay"], g["betas"][0], g["betas"][1], g["eps"]))
        return tuple(out) + tuple(hy) + (ops_generation(),)

    @torch.no_grad()
    def step(self, closure=None, check_tables=True):
        from . import ops
        with ops.use_context(self.context):
            return self._step_in_context(closure, check_tables)

    def _step_in_context(self, closure=None, check_tables=True):
        """check_tables=False skips the (host-side) scan for re-allocated gradients / edited hyper-parameters: use it when the gradients
        live in a fixed flat buffer (lavt_hip.ddp.GradBuckets) and the call is being captured into a hipGraph."""
        loss = closure() if closure is not None else None
        if self._tables is None or (check_tables and self._tables[0] != self._current_key()):
            self._build()
        _, desc, hyper, n, chunks, nchunks, fused = self._tables
        if not check_tables and (self._tables[0][-1] != ops_generation()
                                 or any(p.grad is None or p.grad.data_ptr() != a for p, a in self._probe)):
            # (captured steps skip the host-side scan, but a copy that moved since the table was built would be written at its OLD address, and a
            # gradient buffer that was laid out again -- GradBuckets moves unreported parameters to the late bucket in its second zero() -- would be
            # read at its OLD offsets)
            raise RuntimeError("FusedAdamW.step(check_tables=False): a compute copy was re-allocated or the flat gradient buffer was laid out again "
                               "after the descriptor table was built; call step() once with check_tables=True (outside a capture) first")
        K.check(K.lib.lavt_adamw_step_chunks(K.ptr(desc), K.ptr(hyper), K.ptr(chunks), nchunks, K.ptr(self._step), self.total_steps, self.power, K.stream()))
        # The kernel writes the parameters through raw pointers: p._version does not move, so the cached compute copies (bf16 Linear weights,
        # packed conv weights) are stale now.  They are part of the optimizer's output (fp32 master weights + the compute-dtype copies the next
        # forward reads, as in any mixed-precision trainer): re-cast them here, on the same stream, so that the forward/backward step itself
        # carries no cast kernels (the step harness refreshes them only when asked to: TrainStep(refresh_weights_in_step=True)).
        from . import ops
        ops.weights.refresh_all(done=fused)          # the copies in `fused` were written by the update kernel: only their stamps move
        return loss

    def steps_taken(self) -> int:
        return int(self._step.item()) if self._step is not None else 0

    def current_lr_factor(self) -> float:
        k = self.steps_taken()
        return max(1.0 - k / self.total_steps, 0.0) ** self.power if self.total_steps > 0 else 1.0

    def state_dict(self):
        sd = super().state_dict()
        sd["lavt_schedule"] = {"total_steps": self.total_steps, "power": self.power, "steps_taken": self.steps_taken()}
        return sd

    def load_state_dict(self, state_dict):
        sched = state_dict.get("lavt_schedule")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "lavt_schedule"})
        steps = None
        if sched is not None:
            self.total_steps, self.power, steps = float(sched["total_steps"]), float(sched["power"]), sched["steps_taken"]
        else:
            for st in self.state.values():          # a torch.optim.AdamW checkpoint: per-parameter 'step'
                if "step" in st:
                    steps = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
                    break
        dev = self.param_groups[0]["params"][0].device
        self._step = torch.full((1,), float(steps or 0), dtype=torch.float32, device=dev)
        for st in self.state.values():
            st["step"] = self._step
        self._tables = None
