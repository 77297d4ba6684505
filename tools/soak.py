"""Soak: many training iterations; memory must plateau and losses stay finite (arena pages, table caches, stream joins)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(16, 1, 5, 10, u["C"], u["S"], dev, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time()
for it in range(n):
    tr.do_global_step()
    tr.update_learning_rate()
    gi, di = G.gim_step(trainer, leaked, real, si, defer_join=True)
    if it % 50 == 0 or it == n - 1:
        ops.join_lanes()
        torch.cuda.synchronize()
        print("it %4d  g %.4f d %.4f  alloc %.2f GB reserved %.2f GB  caches: wgq %s pending %d  %.1f s" % (
            it, float(gi[0]), float(di[0]), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            [len(q.cache) for q in ops._QUEUES.values()], len(ops._PENDING_JOIN), time.time() - t0), flush=True)
