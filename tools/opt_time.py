#!/usr/bin/env python3
"""Where the optimizer step's time goes: the multi-tensor AdamW launch vs the refresh of the compute copies (bf16 casts, conv packing, LayerNorm folds)."""
import os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
import lavt_hip
from lavt_hip import ops, _capi as K
from lavt_hip.engine import TrainStep
from lavt_hip.optim import FusedAdamW, lavt_param_groups
from lavt_hip.detweights import det_inputs, fill_state_dict_
from lib import segmentation
DEV = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16)
x, l, m, t = [v.to(DEV) for v in det_inputs(2, 480, 20, seed=3)]
md = segmentation.lavt("", SimpleNamespace(swin_type="base", window12=True, drop_path_rate=0.0)); fill_state_dict_(md); md = md.to(DEV).train()
st = TrainStep(md, x, l, m, t, use_graph=False)
st.warmup_and_capture(eager_iters=1)
opt = FusedAdamW(lavt_param_groups(md), lr=0.0, weight_decay=1e-2, total_steps=1000)
opt.step(); torch.cuda.synchronize()
def timeit(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
_, desc, hyper, n, chunks, nchunks, fused = opt._tables
nparam = sum(p.numel() for g in opt.param_groups for p in g["params"])
print("tensors", n, "parameters %.1f M" % (nparam / 1e6))
t_all = timeit(lambda: opt.step(check_tables=False))
t_adam = timeit(lambda: K.check(K.lib.lavt_adamw_step_chunks(K.ptr(desc), K.ptr(hyper), K.ptr(chunks), nchunks, K.ptr(opt._step), opt.total_steps, opt.power, K.stream())))
t_ref = None
t_ref = timeit(lambda: ops.weights.refresh_all(done=fused))
kinds = {}
for k in ops.weights.store: kinds[k[1:] ] = kinds.get(k[1:], 0) + 1
print("cache entries by kind", kinds)
nb = nparam * 28 + 2 * sum(1 for _ in fused) * 0 + 2 * sum(ops.weights.store[k][1].numel() for k in fused)
print(f"optimizer step {t_all:.3f} ms = adamw {t_adam:.3f} ms + refresh {t_ref:.3f} ms;  chunks {nchunks}, fused copies {len(fused)};  adamw bytes {nb / 1e9:.2f} GB -> {nb / t_adam / 1e6:.0f} GB/s")
