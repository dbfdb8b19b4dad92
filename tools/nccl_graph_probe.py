"""Probe: can RCCL all-reduces (world size 1 here) be captured into a hipGraph through torch.distributed, including a side comm stream?"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
print("nccl", torch.cuda.nccl.version())
x = torch.ones(1 << 20, device=dev); y = torch.ones(1 << 16, device=dev)
comm = torch.cuda.Stream()
def body():
    a = x * 2
    dist.all_reduce(a)                                  # on the current (capture) stream
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        b = y + 1
        dist.all_reduce(b)
    torch.cuda.current_stream().wait_stream(comm)
    return a.sum() + b.sum()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): r = body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print("eager", float(r))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = body()
for _ in range(5): g.replay()
torch.cuda.synchronize()
print("graph", float(r))
dist.destroy_process_group()
print("ok")
