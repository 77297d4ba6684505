"""Run-to-run noise (float atomics + Adam beta1=0) vs. overlapped-vs-sequential difference after 3 iterations."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import episode, filled_sd, load_keys, relerr  # noqa: E402
import optimalstrategiesagainstgenerativeattacks_amd as G  # noqa: E402

tag, cfg = "ovl", "16_1_32"
B, m, n, k, c, s, d = 2, 1, 3, 4, 1, 16, 32
keys = load_keys(cfg)
eps = [[t.float().cuda() for t in episode("%s/%d" % (tag, it), B, m, n, k, c, s, d)] for it in range(3)]


def run(overlap):
    au, im = G.get_au(s, c, d), G.get_im(s, c, d)
    au.load_state_dict(filled_sd(keys["au"], tag + "/au/", torch.float32))
    im.load_state_dict(filled_sd(keys["im"], tag + "/im/", torch.float32))
    au, im = au.cuda(), im.cuda()
    with tempfile.TemporaryDirectory() as td:
        tr = G.GIMImgTrainer(td, m, n, k, au, im, 1e-3, 1e-3, 1e-4, reg_param=0.0)
    trainer = G.DataParallelMock(tr)
    for leaked, real, si, z in eps:
        G.gim_step(trainer, leaked, real, si, z=z, overlap=overlap)
    torch.cuda.synchronize()
    return {k_: v.clone() for k_, v in list(au.state_dict().items()) + list(im.state_dict().items())}


def worst(a, b):
    w = sorted(((relerr(b[k_], a[k_]), k_) for k_ in a), reverse=True)
    return w[:3]


s1, s2, o1, o2 = run(False), run(False), run(True), run(True)
print("seq vs seq ", worst(s1, s2))
print("ovl vs ovl ", worst(o1, o2))
print("seq vs ovl ", worst(s1, o1))
