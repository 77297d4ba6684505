"""Loss-curve deviation against the fp64 oracle over N iterations (tiny config): the oracle in fp32 (= the reference's own
arithmetic), the product on the fp32 MFMA and on the bf16x3 path.   python tools/loss_curve_probe.py [iters]"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimalstrategiesagainstgenerativeattacks_amd as G  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import ops  # noqa: E402
from oracle import gim_oracle as go  # noqa: E402
from tests.helpers import episode, filled_sd, load_keys, relerr  # noqa: E402
from tests.test_gpu_models import _product_models  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
dev = torch.device("cuda:0")
tag, cfg = "curve", "16_1_32"
B, m, n, k, c, s, d = 4, 1, 3, 4, 1, 16, 32
keys = load_keys(cfg)


def f32(sd):
    return {kk: v.float() for kk, v in sd.items()}


o64 = go.OracleTrainer(filled_sd(keys["au"], tag + "/au/"), filled_sd(keys["im"], tag + "/im/"), n, lr, lr, lr * 0.01)
o32 = go.OracleTrainer(f32(filled_sd(keys["au"], tag + "/au/")), f32(filled_sd(keys["im"], tag + "/im/")), n, lr, lr, lr * 0.01)
prods = {}
for name, mode in (("fp32 MFMA", 0), ("bf16x3", 1)):
    au, im = _product_models(tag, cfg)
    with tempfile.TemporaryDirectory() as td:
        tr = G.GIMImgTrainer(td, m, n, k, au, im, lr, lr, lr * 0.01, reg_param=0.0)
    prods[name] = (mode, G.DataParallelMock(tr))
print("%3s | %-23s | %-23s | %-23s   (relative deviation of G loss, D loss from the fp64 oracle)" % ("it", "oracle fp32 (CPU)", "product fp32 MFMA", "product bf16x3"))
for it in range(iters):
    leaked, real, si, z = episode("%s%d" % (tag, it), B, m, n, k, c, s, d)
    g64, d64 = o64.step(leaked, real, si, z)
    g32, d32 = o32.step(leaked.float(), real.float(), si.float(), z.float())
    row = [(relerr(g32[0].mean().double(), g64[0].mean()), relerr(d32[0].mean().double(), d64[0].mean()))]
    for name, (mode, trainer) in prods.items():
        ops.set_conv_precision(mode)
        gi, di = G.gim_step(trainer, *[t.float().to(dev) for t in (leaked, real, si)], z=z.float().to(dev))
        row.append((relerr(gi[0], g64[0].mean()), relerr(di[0], d64[0].mean())))
    ops.set_conv_precision(0)
    print("%3d | %s" % (it, " | ".join("G %.2e  D %.2e" % r for r in row)), flush=True)
