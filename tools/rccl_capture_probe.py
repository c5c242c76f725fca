"""can an RCCL all-reduce be captured into a hipGraph and replayed?  world size 1 on this box's one GPU (the launch path is the same one a
multi-GPU ring takes; what a single rank cannot show is the ring itself).   python tools/rccl_capture_probe.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
dist.all_reduce(t)          # eager warm-up: communicator setup must not happen inside a capture
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(t)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        t.mul_(2.0)
        dist.all_reduce(t)
        t.add_(1.0)
    before = t.clone()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(t, (before * 2 + 1) * 2 + 1)
    print("RCCL all_reduce captured and replayed twice at world size 1: values %s" % ("as expected" if ok else "WRONG"))
except Exception as exc:  # noqa: BLE001
    print("capture of an RCCL all_reduce FAILED: %r" % (exc,))
dist.destroy_process_group()
