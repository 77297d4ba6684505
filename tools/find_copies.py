"""Where do the D2D copies / fills of one generator forward come from?  (torch profiler, grouped by Python call site)"""
import os
import sys
import collections
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
G, tr = bench.build_trainer(u["S"], u["C"], 5, 1, 10, dev)
leaked, real, si = bench.synthetic_batch(16, 1, 5, 10, u["C"], u["S"], dev, 1)
tr.impersonator.train()
for _ in range(2):
    tr.impersonator_sample(leaked)
torch.cuda.synchronize()
sites = collections.Counter()
orig_contig = torch.Tensor.contiguous
orig_clone = torch.Tensor.clone
orig_copy = torch.Tensor.copy_


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "optimalstrategies" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"


def contig(self, *a, **k):
    if not self.is_contiguous():
        sites["contiguous " + site()] += 1
    return orig_contig(self, *a, **k)


def clone(self, *a, **k):
    sites["clone " + site()] += 1
    return orig_clone(self, *a, **k)


def copy_(self, *a, **k):
    sites["copy_ " + site()] += 1
    return orig_copy(self, *a, **k)


torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_ = contig, clone, copy_
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if mode == "fwd":
    tr.impersonator_sample(leaked)
else:
    trainer = G.DataParallelMock(tr)
    G.gim_step(trainer, leaked, real, si, overlap=False)
torch.cuda.synchronize()
torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_ = orig_contig, orig_clone, orig_copy
for k, v in sites.most_common(30):
    print(v, k)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    if mode == "fwd":
        tr.impersonator_sample(leaked)
    else:
        G.gim_step(G.DataParallelMock(tr), leaked, real, si, overlap=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
for ev in prof.key_averages():
    if any(w in ev.key.lower() for w in ("copy", "memcpy", "fill", "zero", "memset")):
        print("%-60s count %d" % (ev.key[:60], ev.count))
# call sites of the copies: stack-enabled second pass
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof2:
    if mode == "fwd":
        tr.impersonator_sample(leaked)
    else:
        G.gim_step(G.DataParallelMock(tr), leaked, real, si, overlap=False)
    torch.cuda.synchronize()
import collections as _c
sites2 = _c.Counter()
for ev in prof2.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::clone", "aten::contiguous", "hipMemcpyAsync", "aten::add", "aten::add_"):
        st = [f for f in (ev.stack or []) if "optimalstrategies" in f]
        sites2[(ev.name, st[0] if st else "?")] += 1
for k, v in sites2.most_common(25):
    print(v, k)
