"""Probe: can an RCCL all-reduce be captured into a hipGraph through torch.distributed on this ROCm / PyTorch?
(world size 1 on one GPU: tells whether ProcessGroupNCCL + hipGraph capture works mechanically)"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        h = dist.all_reduce(y, async_op=True)
        h.wait()
        z = y + 1
    g.replay()
    torch.cuda.synchronize()
    print("NCCL_CAPTURE_OK", float(z[0]))
except Exception as e:  # noqa: BLE001
    print("NCCL_CAPTURE_FAILED", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
