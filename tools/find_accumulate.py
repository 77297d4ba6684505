"""Which parameters still receive their gradient through autograd's AccumulateGrad (an extra add kernel each)?"""
import collections
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
G.gim_step(trainer, leaked, real, si, overlap=False)
hits = collections.Counter()
for mod_name, mod in (("au", tr.authenticator), ("im", tr.impersonator)):
    for name, p in mod.named_parameters():
        p.register_hook(lambda g, nm=mod_name + "." + name: hits.update([nm]) if g is not None else None)
G.gim_step(trainer, leaked, real, si, overlap=False)
torch.cuda.synchronize()
kinds = collections.Counter()
for nm, c in hits.items():
    kinds[".".join(nm.split(".")[-2:]) if not nm.split(".")[-1].startswith("weight_orig") else "weight_orig"] += c
print("parameters with a returned gradient: %d tensors, %d gradient deliveries" % (len(hits), sum(hits.values())))
for k, v in kinds.most_common(20):
    print(v, k)
print(list(hits.items())[:12])
