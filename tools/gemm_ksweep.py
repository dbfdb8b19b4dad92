#!/usr/bin/env python3
"""K sweep of one (M, N) GEMM: separates the fixed cost (launch ramp, prologue latency, epilogue) from the per-K-tile cost.
usage: gemm_ksweep.py [nt|ntk|tn] M N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gemm_bench import timeit, nt, tn
kind, M, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
Ks = [64, 128, 256, 512, 1024, 2048, 4096]
cfgs = [("64/2st", dict(LAVT_GEMM_TILE="64", LAVT_GEMM_STAGES="2")), ("64/3st", dict(LAVT_GEMM_TILE="64", LAVT_GEMM_STAGES="3")),
        ("128/2st/8w", dict(LAVT_GEMM_TILE="128", LAVT_GEMM_STAGES="2")), ("128/3st/8w", dict(LAVT_GEMM_TILE="128", LAVT_GEMM_STAGES="3"))]
if kind == "tn":
    cfgs = [("64 split1", dict(LAVT_TN_SPLIT="1")), ("64 split2", dict(LAVT_TN_SPLIT="2")), ("64 split4", dict(LAVT_TN_SPLIT="4")), ("64 auto", {})]
print(f"{kind} M={M} N={N}: microseconds per launch (TFLOP/s)")
print("%-12s" % "K" + "".join("%16d" % k for k in Ks))
for name, env in cfgs:
    for k in ("LAVT_GEMM_TILE", "LAVT_GEMM_STAGES", "LAVT_TN_SPLIT"): os.environ.pop(k, None)
    os.environ.update(env)
    row = []
    for K in Ks:
        fn = tn(M, N, K) if kind == "tn" else nt(M, N, K, kmajor=(kind == "ntk"))
        t = timeit(fn)
        row.append("%8.1f (%4.0f)" % (t * 1e6, 2.0 * M * N * K / t / 1e12))
    print("%-12s" % name + "".join("%16s" % r for r in row))
