"""Generator forward only (impersonator_sample) - for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(16, 1, 5, 10, u["C"], u["S"], dev, 1)
tr.impersonator.train()
for _ in range(3):
    tr.impersonator_sample(leaked)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    tr.impersonator_sample(leaked)
e1.record()
torch.cuda.synchronize()
print("generator forward: %.2f ms" % (e0.elapsed_time(e1) / 10))
