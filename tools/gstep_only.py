"""Run only the generator step (im_train_step) a few times - for rocprofv3 --kernel-trace --stats of the critical lane."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
m, n, k, B = 1, 5, 10, 16
G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1)
which = sys.argv[1] if len(sys.argv) > 1 else "g"
for _ in range(2):
    G.gim_step(trainer, leaked, real, si, overlap=False)
torch.cuda.synchronize()
fake = G.im_train_step(trainer, leaked, si)[1]
for _ in range(10):
    if which == "g":
        G.im_train_step(trainer, leaked, si)
    else:
        G.au_train_step(trainer, real, fake, si)
torch.cuda.synchronize()
