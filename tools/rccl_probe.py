"""How long does a one-rank RCCL all-reduce of the gradient buckets take (launch with torchrun --nproc-per-node 1)?"""
import os
import time

import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
for n in (61_403_581, 21_826_755, 1_000_000):
    x = torch.randn(n, device=dev)
    for _ in range(3):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for _ in range(10):
        dist.all_reduce(x)
    e1.record()
    torch.cuda.synchronize()
    print("all_reduce %10d floats (%.0f MB): %.3f ms device, %.3f ms host-enqueue+wait per call, world %d"
          % (n, n * 4 / 1e6, e0.elapsed_time(e1) / 10, (time.time() - t0) * 100, dist.get_world_size()), flush=True)
dist.destroy_process_group()
