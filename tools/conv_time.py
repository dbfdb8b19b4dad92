import os, sys
sys.path.insert(0, "lavt-rs_amd")
import torch
from lavt_hip import ops
B, H, W, Cin, Cout = 2, 120, 120, 512, 512
x = torch.randn(B * H * W, Cin, device="cuda:0").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda:0") * (9 * Cin) ** -0.5
with torch.no_grad():
    for _ in range(3): y = ops.conv3x3(x, None, w, B, H, W)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): y = ops.conv3x3(x, None, w, B, H, W)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e3
print(os.environ.get("LAVT_GEMM_WIDE"), os.environ.get("LAVT_GEMM_TILE"), f"{t:.1f} us  {2*B*H*W*Cout*9*Cin/t/1e6:.0f} TF/s")
