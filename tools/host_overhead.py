"""How long does the HOST need to enqueue one training step (no GPU wait)?  If this approaches the device time,
kernel speed-ups stop paying and the step must be captured into a hipGraph."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
for B in (16, 4, 1):
    G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
    trainer = G.DataParallelMock(tr)
    leaked, real, si = bench.synthetic_batch(B, 1, 5, 10, u["C"], u["S"], dev, 1)
    for _ in range(2):
        G.gim_step(trainer, leaked, real, si)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        G.gim_step(trainer, leaked, real, si)
    t_enq = (time.time() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.time() - t0) / n
    print("B=%d: host enqueue %.1f ms/step, wall %.1f ms/step" % (B, t_enq * 1e3, t_all * 1e3), flush=True)
