"""Why does a (no-op, one-rank) RCCL all-reduce inside the training step cost milliseconds?  Host time inside the call vs
step time, for the synchronous call, an async call + later wait, and no call."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
import bench  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import optim  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
u = bench.UNIT["vox64"]
G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(16, 1, 5, 10, u["C"], u["S"], dev, 1)
spent = [0.0]


def make(mode):
    def f(flat_g):
        t0 = time.time()
        if mode == "sync":
            dist.all_reduce(flat_g)
        elif mode == "async":
            w = dist.all_reduce(flat_g, async_op=True)
            w.wait()
        elif mode == "sidestream":
            # collective issued on its own stream; the caller's stream waits with an event only
            cur = torch.cuda.current_stream()
            s = f.stream
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                dist.all_reduce(flat_g)
            cur.wait_stream(s)
        spent[0] += time.time() - t0
        return 1.0
    f.stream = torch.cuda.Stream()
    return f


for mode in ("none", "sync", "async", "sidestream", "none"):
    optim.all_reduce_grads_ = make(mode)
    for _ in range(3):
        G.gim_step(trainer, leaked, real, si, defer_join=True)
    torch.cuda.synchronize()
    spent[0] = 0.0
    t0 = time.time()
    for _ in range(10):
        G.gim_step(trainer, leaked, real, si, defer_join=True)
    torch.cuda.synchronize()
    print("%-10s step %.2f ms, host time inside the 2 collectives per step %.3f ms" % (mode, (time.time() - t0) * 100, spent[0] * 100), flush=True)
dist.destroy_process_group()
