"""Which part of a discriminator step, captured on a forked (non-origin) stream, breaks hipStreamEndCapture?
    python tools/capture_probe.py <case>      (run each case in its own process: a failure is a segfault)"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalstrategiesagainstgenerativeattacks_amd as G  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import ops  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd.gim_img_models import lane_stream  # noqa: E402

case = sys.argv[1]
dev = torch.device("cuda:0")
S, C, D, B, m, n, k = 16, 1, 32, 2, 1, 3, 4
torch.manual_seed(1)
au, im = G.get_au(S, C, D).to(dev), G.get_im(S, C, D).to(dev)
with tempfile.TemporaryDirectory() as td:
    tr = G.GIMImgTrainer(td, m, n, k, au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
trainer = G.DataParallelMock(tr)
mk = lambda t: torch.rand(B, t, C, S, S, device=dev) * 2 - 1  # noqa: E731
leaked, real, si, fake = mk(m), mk(n), mk(k), mk(n)
ds = lane_stream(dev, 1)
for opt in (tr.impersonator_opt, tr.authenticator_opt):
    opt._ensure()


def body():
    cur = torch.cuda.current_stream()
    ds.wait_stream(cur)
    with torch.cuda.stream(ds), ops.lane(1):
        if case == "op":
            y = real * 2
        elif case == "zero":
            tr.authenticator_opt.zero_grad()
        elif case == "adam":
            tr.authenticator_opt.step()
        elif case == "fwd":
            with torch.no_grad():
                out = trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si, grad=False)
        elif case == "fwdg":
            out = trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si)
        elif case == "bwd":
            tr.authenticator_opt.zero_grad()
            out = trainer.forward(mode='authenticator_forward', fake_sample=fake, real_sample=real, si_sample=si)
            out[0].mean().backward()
        elif case == "full":
            G.au_train_step(trainer, real, fake, si)
    cur.wait_stream(ds)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for opt in (tr.impersonator_opt, tr.authenticator_opt):
    opt._ensure()
    opt._push_lrs()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay()
torch.cuda.synchronize()
print("case %s: ok" % case)
