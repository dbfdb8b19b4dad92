#!/usr/bin/env python3
"""Own kernels against the vendor library on the hot GEMM shapes of the Swin-B w12 / batch-2 step, and the box's attainable bf16 rate.

A yardstick, never the product path: `torch.matmul` on ROCm dispatches to hipBLASLt / rocBLAS.  Every row is hipGraph-timed (20 launches per
replay, 5 replays; no host time between kernels) on random bf16 operands.  Columns: own kernel as the product dispatches it (plain epilogue),
vendor GEMM of the same shape, ratio, and both as a fraction of the nominal 2.5 PFLOP/s.  Written for profiles/r05_gemm_yardstick.txt.

    python tools/gemm_yardstick.py > gpurun_out/r05_gemm_yardstick.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lavt-rs_amd"))
import torch  # noqa: E402
from lavt_hip import ops  # noqa: E402

dev = "cuda:0"
bf = torch.bfloat16
PEAK = 2500.0


def timeit(fn, iters=20, reps=5):
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
    own = lambda: ops.gemm_nt(bf, M, N, K, A, K, B, N if kmajor else K, C, N, b_kmajor=kmajor)          # noqa: E731
    Bv = B if kmajor else B.t()
    out = torch.empty(M, N, device=dev, dtype=bf)
    ven = lambda: torch.matmul(A, Bv, out=out)          # noqa: E731
    return own, ven


def tn(I, J, K):
    A = torch.randn(K, I, device=dev).to(bf)
    B = torch.randn(K, J, device=dev).to(bf)
    C = torch.zeros(I, J, device=dev)
    own = lambda: ops.gemm_tn(bf, I, J, K, A, I, B, J, C, J)          # noqa: E731  (fp32 gradient, accumulated)
    At = A.t()
    try:
        torch.mm(At, B, out_dtype=torch.float32)
        ven = lambda: torch.mm(At, B, out_dtype=torch.float32)          # noqa: E731
    except Exception:  # noqa: BLE001
        ven = lambda: torch.mm(At, B)          # noqa: E731  (bf16 output: fewer bytes written than the fp32 gradient)
    return own, ven


# (label, kind, a, b, c, launches per step of this shape class in the headline workload)
SHAPES = [
    ("fc1 s2 fwd", "nt", 1800, 2048, 512, 18), ("fc2 s2 fwd", "nt", 1800, 512, 2048, 18), ("proj s2 fwd", "nt", 1800, 512, 512, 18),
    ("qkv s2 fwd (unfused)", "nt", 2592, 1536, 512, 0),
    ("d-fc2 s2", "ntk", 1800, 2048, 512, 18), ("d-fc1 s2", "ntk", 1800, 512, 2048, 18), ("d-qkv s2", "ntk", 1800, 512, 1536, 18), ("d-proj s2", "ntk", 1800, 512, 512, 18),
    ("w-fc1 s2", "tn", 2048, 512, 1800, 18), ("w-fc2 s2", "tn", 512, 2048, 1800, 18), ("w-qkv s2", "tn", 1536, 512, 1800, 18), ("w-proj s2", "tn", 512, 512, 1800, 18),
    ("fc1 s1 fwd", "nt", 7200, 1024, 256, 2), ("fc1 s0 fwd", "nt", 28800, 512, 128, 2), ("fc1 s3 fwd", "nt", 450, 4096, 1024, 2), ("fc2 s3 fwd", "nt", 450, 1024, 4096, 2),
    ("conv2_2 as GEMM", "nt", 28800, 512, 4608, 2), ("conv @60 as GEMM", "nt", 7200, 512, 4608, 2),
    ("square 4096", "nt", 4096, 4096, 4096, 0), ("square 8192 (attainable peak)", "nt", 8192, 8192, 8192, 0),
]


# the same layer shapes at batch 4 (3600 rows) and in the video workload's stage 2 (4608 rows): the rows the ring-depth rule of the second session of round 5 changed
# (`python tools/gemm_yardstick.py b4`; LAVT_PROBE=0,0,0,0,0,0,0,600 gives the old rule)
SHAPES_B4 = [
    ("fc1 s2 fwd b4", "nt", 3600, 2048, 512, 18), ("fc2 s2 fwd b4", "nt", 3600, 512, 2048, 18), ("proj s2 fwd b4", "nt", 3600, 512, 512, 18),
    ("d-fc2 s2 b4", "ntk", 3600, 2048, 512, 18), ("d-fc1 s2 b4", "ntk", 3600, 512, 2048, 18), ("d-qkv s2 b4", "ntk", 3600, 512, 1536, 18),
    ("fc1 s2 fwd video", "nt", 4608, 2048, 512, 18), ("fc2 s2 fwd video", "nt", 4608, 512, 2048, 18), ("d-fc2 s2 video", "ntk", 4608, 2048, 512, 18), ("qkv s2 video (unfused)", "nt", 6272, 1536, 512, 18),
    ("fc1 s1 fwd b4", "nt", 14400, 1024, 256, 2), ("fc1 s0 fwd b4", "nt", 57600, 512, 128, 2),
]


def main():
    global SHAPES
    if len(sys.argv) > 1 and sys.argv[1] == "b4":
        SHAPES = SHAPES_B4
    print(f"# {torch.cuda.get_device_name(0)}; torch {torch.__version__}; nominal bf16 dense peak {PEAK:.0f} TFLOP/s")
    print(f"# {'shape':32s} {'kind':4s} {'M/I':>6s} {'N/J':>6s} {'K':>6s} | {'own us':>8s} {'TF/s':>7s} {'of peak':>7s} | {'vendor us':>9s} {'TF/s':>7s} {'of peak':>7s} | own/vendor time")
    tot_own = tot_ven = 0.0
    for name, kind, a, b, c, n in SHAPES:
        own, ven = nt(a, b, c, kind == "ntk") if kind != "tn" else tn(a, b, c)
        it = 5 if 2.0 * a * b * c > 5e10 else 20
        to, tv = timeit(own, iters=it), timeit(ven, iters=it)
        fl = 2.0 * a * b * c
        print(f"  {name:32s} {kind:4s} {a:6d} {b:6d} {c:6d} | {to * 1e6:8.1f} {fl / to / 1e12:7.1f} {fl / to / 1e12 / PEAK:7.3f} | {tv * 1e6:9.1f} {fl / tv / 1e12:7.1f} {fl / tv / 1e12 / PEAK:7.3f} | {to / tv:5.2f}")
        tot_own += n * to
        tot_ven += n * tv
        del own, ven
        torch.cuda.empty_cache()
    print(f"# launches-per-step-weighted sum over the rows above: own {tot_own * 1e3:.3f} ms, vendor {tot_ven * 1e3:.3f} ms per step")
    # the four stage-2 weight gradients as the product runs them: ONE grouped launch (tools/wgrad_group_time.py times it with its riders)


if __name__ == "__main__":
    main()
