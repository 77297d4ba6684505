"""Error of the conv kernels against an fp64 reference (torch CPU), fp32-MFMA path vs bf16x3 path, on benchmark layer shapes:
forward, dgrad and wgrad through the autograd operator (ops.conv2d), with zero-mean data and with all-positive data (sums of
same-sign terms are the worst case for the bf16 MFMA's truncating accumulate).   python tools/bf16x3_accuracy.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [(4, 64, 64, 3, 64, False), (4, 128, 256, 3, 16, False), (4, 512, 512, 3, 8, False), (4, 64, 64, 3, 64, True), (4, 256, 256, 3, 16, True),
          (2, 64, 64, 9, 32, False)]   # N, Cin, Cout, K, H, pool-fold


def err(a, ref):
    a = a.double().cpu()
    return float((a - ref).abs().max() / ref.abs().max()), float(((a - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


print("%-34s %-9s | %-25s | %-25s" % ("shape (N,Cin,Cout,K,H,pool)", "data", "fp32 MFMA  max / rms", "bf16x3     max / rms"))
for (N, Cin, Cout, K, H, pool) in SHAPES:
    for kind in ("zero-mean", "positive"):
        g = torch.Generator().manual_seed(7)
        x = torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, Cin, K, K, generator=g, dtype=torch.float64) / (Cin * K * K) ** 0.5
        r = torch.randn(N, Cout, H >> pool, H >> pool, generator=g, dtype=torch.float64)
        if kind == "positive":
            x, w, r = x.abs(), w.abs(), r.abs()
        x.requires_grad_()
        w.requires_grad_()
        y = F.conv2d(x, w, None, padding=(K - 1) // 2)
        if pool:
            y = F.avg_pool2d(y, 2)
        (y * r).sum().backward()
        res = {}
        for mode in (0, 1):
            ops.set_conv_precision(mode)
            xg = x.detach().permute(0, 2, 3, 1).contiguous().float().to(dev).requires_grad_()
            wg = w.detach().float().to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            yg = ops.conv2d(xg, wg, None, None, None, None, None, 0, 1.0, pool=bool(pool))
            (yg * r.permute(0, 2, 3, 1).contiguous().float().to(dev)).sum().backward()
            res[mode] = (err(yg.detach().permute(0, 3, 1, 2), y.detach()), err(xg.grad.permute(0, 3, 1, 2), x.grad), err(wg.grad, w.grad))
        ops.set_conv_precision(0)
        for i, name in enumerate(("fwd", "dgrad", "wgrad")):
            print("%-34s %-9s | %-5s %.2e / %.2e     | %.2e / %.2e" % ((N, Cin, Cout, K, H, int(pool)), kind, name, res[0][i][0], res[0][i][1], res[1][i][0], res[1][i][1]), flush=True)
