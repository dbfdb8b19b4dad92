#!/usr/bin/env python3
"""Where do the step harness and plain autograd differ?  Gradients of the Swin-T micro model three ways: plain eager autograd (twice: run-to-run
noise), the harness run eagerly, the harness replayed from its hipGraph."""
import os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch, torch.nn.functional as F
import lavt_hip
from lavt_hip import ops
from lavt_hip.engine import TrainStep
from lavt_hip.detweights import det_inputs, fill_state_dict_
from lib import segmentation
DEV = "cuda:0"
lavt_hip.set_compute_dtype(torch.bfloat16)
x, l, m, t = [v.to(DEV) for v in det_inputs(2, 96, 20, seed=3)]
def build():
    md = segmentation.lavt("", SimpleNamespace(swin_type="tiny", drop_path_rate=0.0)); fill_state_dict_(md); return md.to(DEV).train()
def plain():
    md = build()
    loss = F.cross_entropy(md(x, l, m), t, weight=torch.tensor([0.9, 1.1], device=DEV)); loss.backward()
    print("plain loss %.8f" % float(loss))
    return {n: p.grad.clone() for n, p in md.named_parameters() if p.grad is not None}
def harness(graph, fused_loss=True, refresh=True):
    md = build()
    st = TrainStep(md, x, l, m, t, use_graph=graph, fused_loss=fused_loss)
    if not refresh:
        ops.weights.refresh_all = lambda: None
        ops.weights.build_multicast = lambda d: None
    st.warmup_and_capture(eager_iters=1)
    st.step(); torch.cuda.synchronize()
    print("harness loss %.8f (graph %s fused_loss %s refresh %s)" % (float(st.loss), graph, fused_loss, refresh))
    out = {n: p.grad.clone() for n, p in md.named_parameters() if p.grad is not None}
    ops.sinks.clear(); ops.wgrads.enabled = False
    return out
def diff(a, b, tag):
    rows = []
    for n in a:
        if n in b:
            sc = float(a[n].abs().max())
            rows.append((float((a[n] - b[n]).abs().max()) / max(sc, 1e-12), n, sc))
    rows.sort(reverse=True)
    print(f"--- {tag}: worst {rows[0][0]:.2e}")
    for r in rows[:int(os.environ.get('TOP', 6))]:
        print(f"   {r[0]:.3e}  scale {r[2]:.2e}  {r[1]}")
A, A2 = plain(), plain()
diff(A, A2, "plain vs plain")
B = harness(False)
diff(A, B, "plain vs harness (eager)")
C = harness(True)
diff(B, C, "harness eager vs harness graph")
D = harness(False, fused_loss=False)
diff(A, D, "plain vs harness (eager, plain loss)")
E = harness(False, refresh=False)
diff(A, E, "plain vs harness (eager, no refresh_all)")
