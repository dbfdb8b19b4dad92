#!/usr/bin/env python3
"""Pipeline-depth sweep of the NT GEMM at the stage-2 shapes of Swin-B (batch 2): tile x stages."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gemm_bench import timeit, nt
SH = [("qkv", "nt", 2592, 1536, 512), ("proj", "nt", 2592, 512, 512), ("fc1", "nt", 1800, 2048, 512), ("fc2", "nt", 1800, 512, 2048),
      ("d-qkv", "ntk", 2592, 512, 1536), ("d-proj", "ntk", 2592, 512, 512), ("d-fc1", "ntk", 1800, 512, 2048), ("d-fc2", "ntk", 1800, 2048, 512),
      ("conv2_2", "nt", 28800, 512, 4608)]
for name, kind, M, N, K in SH:
    res = []
    for tile in ("64", "128"):
        for st in ("2", "3", "4"):
            os.environ.update(LAVT_GEMM_TILE=tile, LAVT_GEMM_STAGES=st)
            t = timeit(nt(M, N, K, kmajor=(kind == "ntk")))
            res.append(f"t{tile}s{st}: {t*1e6:5.1f}")
    os.environ.pop("LAVT_GEMM_TILE"); os.environ.pop("LAVT_GEMM_STAGES")
    t = timeit(nt(M, N, K, kmajor=(kind == "ntk")))
    print(f"{name:8s} {M:6d} {N:5d} {K:5d} | " + " | ".join(res) + f" | auto: {t*1e6:5.1f}")
