"""Which torch (ATen) operators launch kernels of their own inside one training iteration - the fills, adds and copies that are
not this package's HIP kernels - with counts per step and the Python frames that issue them.
    python tools/torch_op_census.py [episodes=16] [reg_param=0]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
m, n, k = 1, 5, 10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REG = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev, reg_param=REG)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1)
for _ in range(3):
    G.gim_step(trainer, leaked, real, si)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        G.gim_step(trainer, leaked, real, si)
    torch.cuda.synchronize()
ops = collections.Counter()
stacks = collections.defaultdict(collections.Counter)
shapes = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if not any(kk.name for kk in ev.kernels):
        continue
    # only leaf aten ops that launched kernels themselves
    ops[ev.name] += 1
    fr = [s for s in (ev.stack or []) if "optimalstrategies" in s or "bench.py" in s]
    stacks[ev.name][" <- ".join(f.split("/")[-1] for f in fr[:3]) or "(autograd engine / torch internals)"] += 1
    shapes[ev.name][str(ev.input_shapes)[:80]] += 1
allops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::") or "Memcpy" in ev.name or "Memset" in ev.name)
print("all ATen operator / memcpy / memset events per step: " + ", ".join("%s %.0f" % (nm, c / STEPS) for nm, c in allops.most_common(40)))
for nm in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zero_", "aten::zeros", "aten::empty_like"):
    cs = collections.Counter()
    for ev in prof.events():
        if ev.name == nm:
            fr = [s_ for s_ in (ev.stack or []) if "optimalstrategies" in s_ or "bench.py" in s_]
            cs[" <- ".join(f.split("/")[-1] for f in fr[:3]) or "(autograd engine / torch internals)"] += 1
    print(nm, ["%.0f %s" % (c / STEPS, k_) for k_, c in cs.most_common(6)])
print("ATen operators that launched kernels, per step (of %d profiled steps):" % STEPS)
for name, c in ops.most_common(25):
    print("%-28s %6.1f / step" % (name, c / STEPS))
    for stck, cc in stacks[name].most_common(6):
        print("      %6.1f  %s" % (cc / STEPS, stck))
    for shp, cc in shapes[name].most_common(4):
        print("      %6.1f  shapes %s" % (cc / STEPS, shp))
