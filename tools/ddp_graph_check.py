"""Dev check: TrainStep eager vs hipGraph in a 1-rank RCCL group (forced collectives): which gradients differ."""
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("LAVT_FORCE_COLLECTIVES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lavt-rs_amd"), ROOT]
import lavt_hip
from types import SimpleNamespace
from lavt_hip.detweights import det_inputs, fill_state_dict_
from lavt_hip.engine import TrainStep
from lib import segmentation
torch.cuda.set_device(0)
force = os.environ["LAVT_FORCE_COLLECTIVES"] == "1"
if force:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lavt_hip.set_compute_dtype(torch.bfloat16)
if os.environ.get("CHECK_NOAR") == "1":
    class _W:
        def wait(self): pass
    dist.all_reduce = lambda *a, **k: _W()
res = []
import threading
from lavt_hip import ddp as _ddp, ops as _ops
_orig_launch = _ddp.GradBuckets._launch
def _dbg_launch(self, b):
    if os.environ.get("CHECK_TRACE") == "1":
        print("LAUNCH bucket", b, "thread", threading.get_ident() == threading.main_thread().ident, "cur stream", hex(torch.cuda.current_stream().cuda_stream), "pending", self.pending, "expected", self.expected, flush=True)
    return _orig_launch(self, b)
_ddp.GradBuckets._launch = _dbg_launch
_orig_on = _ddp.GradBuckets._on_grad
_fires = {}
import traceback
_traced = [0]
def _dbg_on(self, p):
    _fires[id(p)] = _fires.get(id(p), 0) + 1
    if _traced[0] < 2 and p.dim() == 2 and p.shape[0] == 2304:      # a stage-3 qkv weight (Swin-T: 768 -> 2304)
        _traced[0] += 1
        print("FIRE", _fires[id(p)], "".join(traceback.format_stack(limit=6)), flush=True)
    return _orig_on(self, p)
if os.environ.get("CHECK_FIRES") == "1":
    _ddp.GradBuckets._on_grad = _dbg_on
_real_ar = dist.all_reduce
class _W:
    def wait(self): pass
for it_no, use_graph in enumerate((False, True, False)):
    dist.all_reduce = (lambda *a, **k: _W()) if (it_no == 0 and os.environ.get("CHECK_BASE") == "1") else _real_ar
    model = segmentation.lavt("", SimpleNamespace(swin_type="tiny", drop_path_rate=0.0))
    fill_state_dict_(model)
    model = model.cuda()
    if force and os.environ.get('CHECK_SYNCBN', '1') == '1':
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model.train()
    x, l, m, t = det_inputs(2, 96, 20, seed=3)
    step = TrainStep(model, x.cuda(), l.cuda(), m.cuda(), t.cuda(), world=2 if force else 1, use_graph=use_graph, bucket_mib=float(os.environ.get('CHECK_BUCKET', '64')))
    if os.environ.get("CHECK_FIRES") == "1":
        _fires.clear()
        step._body(); torch.cuda.synchronize()
        names_by_id = {id(p): n for n, p in model.named_parameters()}
        multi = sorted((c, names_by_id.get(i, "?")) for i, c in _fires.items() if c != 1)
        print("FIRES != 1:", len(multi), multi[:40], flush=True)
        sys.exit(0)
    step.warmup_and_capture(eager_iters=2)
    step.step(); step.step()
    torch.cuda.synchronize()
    names = [n for n, p in reversed(list(model.named_parameters())) if p.requires_grad]
    res.append((float(step.loss), step.buckets.flat.clone(), names, [p.numel() for p in reversed([p for p in model.parameters() if p.requires_grad])], step.captured))
for a, b in ((0, 1), (0, 2)):
    g0, g1 = res[a][1], res[b][1]
    print("pair", a, b, "captured", res[a][4], res[b][4], "loss", res[a][0], res[b][0], "max rel", float((g0 - g1).abs().max() / g0.abs().max()))
    off = 0
    bad = []
    for n, k in zip(res[0][2], res[0][3]):
        d = float((g0[off:off + k] - g1[off:off + k]).abs().max()); s = float(g0[off:off + k].abs().max())
        if d > 1e-4 * max(s, 1e-8):
            a_, b_ = g0[off:off + k], g1[off:off + k]
            bad.append((d / max(s, 1e-8), n, "ratio(b/a) at max: %.3f" % float((b_ / a_)[(a_ - b_).abs().argmax()]), "frac differing %.3f" % float(((a_ - b_).abs() > 1e-4 * s).float().mean())))
        off += k
    bad.sort(reverse=True)
    print("   differing tensors:", len(bad), bad[:8])
if force:
    dist.destroy_process_group()
