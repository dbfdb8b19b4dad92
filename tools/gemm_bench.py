#!/usr/bin/env python3
"""Micro-benchmark of the gather-GEMM families on the shapes of the Swin-B w12 / batch-2 step (run on the GPU box).
Reports TFLOP/s per tile configuration (LAVT_GEMM_TILE) so the dispatch heuristic in csrc/gemm.hip can be tuned."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch  # noqa: E402
from lavt_hip import ops, _capi as K  # noqa: E402

dev = "cuda:0"
bf = torch.bfloat16


def timeit(fn, iters=20, reps=5):
    """GPU time per launch: the launches are captured into a hipGraph (no Python/ctypes time between kernels)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e-3


def nt(M, N, K, kmajor=False):
    A = torch.randn(M, K, device=dev).to(bf)
    B = torch.randn((K, N) if kmajor else (N, K), device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    return lambda: ops.gemm_nt(bf, M, N, K, A, K, B, N if kmajor else K, C, N, b_kmajor=kmajor)


def tn(I, J, K):
    A = torch.randn(K, I, device=dev).to(bf)
    B = torch.randn(K, J, device=dev).to(bf)
    C = torch.zeros(I, J, device=dev)
    return lambda: ops.gemm_tn(bf, I, J, K, A, I, B, J, C, J)


SHAPES = [
    ("conv2_2", "nt", 28800, 512, 4608), ("conv1_2-ish", "nt", 28800, 512, 5760), ("conv @60", "nt", 7200, 512, 4608),
    ("qkv s2", "nt", 2592, 1536, 512), ("proj s2", "nt", 2592, 512, 512), ("fc1 s2", "nt", 1800, 2048, 512), ("fc2 s2", "nt", 1800, 512, 2048),
    ("qkv s0", "nt", 28800, 384, 128), ("fc1 s0", "nt", 28800, 512, 128), ("fc2 s0", "nt", 28800, 128, 512), ("fc1 s1", "nt", 7200, 1024, 256),
    ("fc1 s3", "nt", 450, 4096, 1024), ("fc2 s3", "nt", 450, 1024, 4096),
    ("pwam s0", "nt", 28800, 128, 128), ("pwam s1", "nt", 7200, 256, 256), ("pwam s2", "nt", 1800, 512, 512), ("d-pwam s0", "ntk", 28800, 128, 128),
    ("d-qkv s2", "ntk", 2592, 512, 1536), ("d-fc1 s2", "ntk", 1800, 512, 2048), ("d-fc2 s2", "ntk", 1800, 2048, 512), ("d-fc2 s0", "ntk", 28800, 512, 128),
    ("w-conv2_2", "tn", 512, 4608, 28800), ("w-qkv s2", "tn", 1536, 512, 2592), ("w-fc1 s2", "tn", 2048, 512, 1800), ("w-fc2 s2", "tn", 512, 2048, 1800), ("w-fc1 s0", "tn", 512, 128, 28800),
    ("w-qkv s0", "tn", 384, 128, 28800), ("w-fc1 s3", "tn", 4096, 1024, 450), ("w-pwam s0", "tn", 128, 128, 28800),
]

def main():
    for name, kind, a, b, c in SHAPES:
        if len(sys.argv) > 1 and kind != sys.argv[1] and sys.argv[1] not in name:
            continue
        res = []
        if kind == "tn":
            for v2 in ("0", "1"):
                for tile in ("64", "128"):
                    for sp in ("1", "2", "4", "8"):
                        os.environ.update(LAVT_GEMM_TILE=tile, LAVT_TN_SPLIT=sp, LAVT_GEMM_V2=v2)
                        K.lib.lavt_tuning_reload()          # the library reads its switches once per process
                        t = timeit(tn(a, b, c), iters=5 if c > 20000 else 20)
                        res.append((f"v{int(v2)+1}t{tile}/s{sp}", t * 1e6, 2.0 * a * b * c / t / 1e12))
            for k in ("LAVT_GEMM_TILE", "LAVT_TN_SPLIT", "LAVT_GEMM_V2"):
                os.environ.pop(k)
            K.lib.lavt_tuning_reload()
            print(f"{name:9s} {kind:3s} {a:6d} {b:5d} {c:5d} | " + " | ".join(f"{k}: {us:5.1f}us" for k, us, tf in res))
            continue
        for tile, stages, v2, wv in (("64", "3", "0", "4"), ("64", "2", "1", "4"), ("64", "3", "1", "4"), ("128", "2", "1", "4"), ("128", "2", "1", "8"), ("128", "3", "1", "8"), ("256", "2", "1", "8"), ("512", "2", "1", "16")):
            os.environ["LAVT_GEMM_TILE"] = tile
            os.environ["LAVT_GEMM_STAGES"] = stages
            os.environ["LAVT_GEMM_V2"] = v2
            os.environ["LAVT_GEMM_WAVES"] = wv
            K.lib.lavt_tuning_reload()
            fn = nt(a, b, c) if kind == "nt" else nt(a, b, c, True) if kind == "ntk" else tn(a, b, c)
            t = timeit(fn)
            res.append((("v2" if v2 == "1" else "v1") + "-" + tile + ("s" + stages + "w" + wv if v2 == "1" else ""), t * 1e6, 2.0 * a * b * c / t / 1e12))
        for k in ("LAVT_GEMM_TILE", "LAVT_GEMM_STAGES", "LAVT_GEMM_V2", "LAVT_GEMM_WAVES"):
            os.environ.pop(k)
        K.lib.lavt_tuning_reload()
        fn = nt(a, b, c) if kind == "nt" else nt(a, b, c, True) if kind == "ntk" else tn(a, b, c)
        t = timeit(fn)
        # yardstick only (never on the product path): the vendor library's GEMM on the same shape through torch.matmul (hipBLASLt)
        Am = torch.randn(a, c, device=dev).to(bf)
        Bm = torch.randn((c, b) if kind == "ntk" else (b, c), device=dev).to(bf)
        tl = timeit((lambda: torch.matmul(Am, Bm)) if kind == "ntk" else (lambda: torch.matmul(Am, Bm.t())))
        print(f"{name:9s} {kind:3s} {a:6d} {b:5d} {c:5d} | " + " | ".join(f"{tile}: {us:5.1f}us {tf:4.0f}" for tile, us, tf in res) + f" | auto: {t * 1e6:5.1f}us | hipBLASLt: {tl * 1e6:5.1f}us")


if __name__ == "__main__":
    main()
