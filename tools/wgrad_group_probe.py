#!/usr/bin/env python3
"""The four weight gradients of a stage-2 Swin-B block as the step issues them (qkv: B rows through the window map, proj: A rows through it,
fc1 / fc2 plain), as ONE grouped launch vs split into the mapped pair and the plain pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch
from lavt_hip import ops, rowmaps
from lavt_hip import _capi as K
dev, bf = "cuda:0", torch.bfloat16


def timeit(fn, iters=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e-3


wm = rowmaps.window_map(2, 30, 30, 12, 6, torch.device(dev))        # [2592] token row (or -1) of each windowed row
T, Wr, C = 1800, wm.numel(), 512
keep = []


class Q:
    def __init__(self): self.items = []
    def add(self, p, t): self.items.append(p); keep.append(t)


def prob(I, J, Krows, a_rows, b_rows, a_map=None, b_map=None):
    A = torch.randn(a_rows, I, device=dev).to(bf); B = torch.randn(b_rows, J, device=dev).to(bf); Cc = torch.zeros(I, J, device=dev)
    q = Q()
    ops.gemm_tn(bf, I, J, Krows, A, I, B, J, Cc, J, a_rowmap=a_map, b_rowmap=b_map, defer=q)
    return q.items[0]


qkv = prob(1536, 512, Wr, Wr, T, b_map=wm)          # dqkv windowed rows x LN1(x) token rows gathered
proj = prob(512, 512, Wr, T, Wr, a_map=wm)          # dy token rows gathered x attention output windowed rows
fc1 = prob(2048, 512, T, T, T)
fc2 = prob(512, 2048, T, T, T)


def launch(ps):
    arr = (K.GemmTN * len(ps))(*ps)
    keep.append(arr)
    return lambda: K.check(K.lib.lavt_gemm_tn_grouped(arr, len(ps), K.stream()))


all4, mapped, plain = launch([qkv, proj, fc1, fc2]), launch([qkv, proj]), launch([fc1, fc2])
both = lambda: (mapped(), plain())
print(f"one launch of 4      : {timeit(all4)*1e6:6.1f} us")
print(f"mapped pair          : {timeit(mapped)*1e6:6.1f} us")
print(f"plain pair           : {timeit(plain)*1e6:6.1f} us")
print(f"mapped + plain       : {timeit(both)*1e6:6.1f} us")
