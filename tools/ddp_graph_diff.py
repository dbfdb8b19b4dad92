#!/usr/bin/env python3
"""Eager vs captured step in a 1-rank RCCL group (LAVT_FORCE_COLLECTIVES): which parameters' gradients differ, and which bucket they sit in."""
import os, sys
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400), RANK="0", WORLD_SIZE="1", LAVT_FORCE_COLLECTIVES="1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lavt-rs_amd"), ROOT]
import torch, torch.distributed as dist
import lavt_hip
from types import SimpleNamespace
from lavt_hip.detweights import det_inputs, fill_state_dict_
from lavt_hip.engine import TrainStep
from lib import segmentation
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lavt_hip.set_compute_dtype(torch.bfloat16)
res = []
for use_graph in (False, True):
    model = segmentation.lavt("", SimpleNamespace(swin_type="tiny", drop_path_rate=0.0))
    fill_state_dict_(model)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model.cuda()).train()
    x, l, m, t = det_inputs(2, 96, 20, seed=3)
    step = TrainStep(model, x.cuda(), l.cuda(), m.cuda(), t.cuda(), world=2, use_graph=use_graph)
    step.warmup_and_capture(eager_iters=2)
    step.step(); step.step()
    torch.cuda.synchronize()
    res.append((step, model, step.buckets.flat.clone()))
(s0, m0, g0), (s1, m1, g1) = res
scale = float(g0.abs().max())
off = 0
names = {id(p): n for n, p in m0.named_parameters()}
print("buckets", s0.buckets.buckets)
for p in reversed(s0.buckets.params):
    n = p.numel()
    e = float((g0[off:off + n] - g1[off:off + n]).abs().max()) / scale
    if e > 1e-4:
        print(f"{names[id(p)]:70s} bucket {s0.buckets.bucket_of[p]} off {off} n {n} err {e:.3e}")
    off += n
dist.destroy_process_group()
