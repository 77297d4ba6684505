"""Where does the HOST spend the ~40 ms it needs to enqueue one training step?  cProfile of gim_step at one episode (the GPU then
never limits), autograd worker thread disabled so that the backward's Python shows up in the same profile.
    python tools/host_profile.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(1, 1, 5, 10, u["C"], u["S"], dev, 1)
for _ in range(3):
    G.gim_step(trainer, leaked, real, si, defer_join=True)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    G.gim_step(trainer, leaked, real, si, defer_join=True)
print("multi-threaded autograd: %.1f ms/step host" % ((time.time() - t0) / n * 1e3))
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
for _ in range(2):
    G.gim_step(trainer, leaked, real, si, defer_join=True)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    G.gim_step(trainer, leaked, real, si, defer_join=True)
print("single-threaded autograd: %.1f ms/step host" % ((time.time() - t0) / n * 1e3))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    G.gim_step(trainer, leaked, real, si, defer_join=True)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(70)
