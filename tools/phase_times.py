"""Where one training iteration spends its GPU time when nothing overlaps: generator forward, its discriminator forward,
generator backward (+Adam), discriminator-step forward, backward (+Adam).  Sequential protocol, HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
u = bench.UNIT["vox64"]
m, n, k, B = 1, 5, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 16
G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
trainer = G.DataParallelMock(tr)
leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1)
for _ in range(3):
    G.gim_step(trainer, leaked, real, si, overlap=False)
torch.cuda.synchronize()


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


acc = [0.0] * 5
R = 5
for _ in range(R):
    tr.impersonator.train(); tr.impersonator_opt.zero_grad()
    e0 = ev()
    fake = tr.impersonator(leaked_sample=leaked, n=n)
    e1 = ev()
    out = tr.authenticator(test_sample=fake, si_sample=si)
    loss = tr.gan_loss(out, 1.0).mean()
    e2 = ev()
    loss.backward(); tr.impersonator_opt.step()
    e3 = ev()
    tr.authenticator.train(); tr.authenticator_opt.zero_grad()
    res = tr.authenticator_forward(fake.detach(), real, si)
    dl = res[0].mean()
    e4 = ev()
    dl.backward(); tr.authenticator_opt.step()
    e5 = ev()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(((e0, e1), (e1, e2), (e2, e3), (e3, e4), (e4, e5))):
        acc[i] += a.elapsed_time(b) / R
names = ["G nets forward", "G step: D forward (fake, si) [D weights NOT frozen here: includes nothing extra in fwd]",
         "G backward (through D) + Adam", "D step forward (si, real, fake)", "D step backward + Adam"]
for nm, t in zip(names, acc):
    print("%-70s %7.2f ms" % (nm, t))
print("sum %.2f ms" % sum(acc))
