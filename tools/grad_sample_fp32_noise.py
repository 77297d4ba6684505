"""How far the REFERENCE'S OWN arithmetic in fp32 (the oracle run in float32 on the CPU) is from its fp64 run on the
gradient-tensor samples of the benchmark-shape fixtures (tests/golden/nets_<tag>_grads.npz): the noise floor of the
position-sensitive gradient check in tests/test_gpu_models.py::_check_grad_samples.   python tools/grad_sample_fp32_noise.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gim_oracle as go  # noqa: E402
from tests.helpers import episode, filled_sd, load_json, load_keys, load_npz, relerr  # noqa: E402

torch.set_num_threads(8)
for tag, cfg in (("om32_f64", "32_1_512"), ("vox64_f64", "64_3_512")):
    gs = load_npz("nets_%s_grads.npz" % tag)
    c = load_json("nets_%s.json" % tag)["config"]
    keys = load_keys(cfg)
    for seed_note, perturb in (("fp32", 0.0),):
        au = filled_sd(keys["au"], tag + "/au/", torch.float32)
        im = filled_sd(keys["im"], tag + "/im/", torch.float32)
        go.set_requires_grad(au)
        go.set_requires_grad(im)
        leaked, real, si, z = episode(tag, c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"], torch.float32)
        loss, fake, out = go.impersonator_forward(au, im, leaked, si, c["n"], z, True, True)
        loss.mean().backward()
        rows, allrows = [], []

        def all_rows(what, prefix, sd):
            # (round 3) thin samples of EVERY parameter gradient: error relative to ||ref|| + 1e-4 of the largest sample norm of the
            # pass (gradients that are mathematically zero - a conv bias in front of a norm layer - are rounding noise on both sides)
            floor = 1e-4 * max(float(torch.as_tensor(gs[k]).double().norm()) for k in gs.files if k.startswith(prefix))
            for k in gs.files:
                if k.startswith(prefix):
                    _, stride, name = k.split("/", 2)
                    ref = torch.as_tensor(gs[k]).double()
                    got = sd[name].grad.double().reshape(-1)[::int(stride)]
                    allrows.append((what, name, float((got - ref).norm() / (ref.norm() + floor))))
        for k in gs.files:
            if k.startswith("g/"):
                _, stride, name = k.split("/", 2)
                rows.append(("G step", name, relerr(im[name].grad.double().reshape(-1)[::int(stride)], gs[k])))
        all_rows("G step", "g_all/", im)
        for sd in (au, im):
            for p in sd.values():
                p.grad = None
        go.authenticator_forward(au, fake.detach(), real, si, True, 0.0)[0].mean().backward()
        for k in gs.files:
            if k.startswith("d/"):
                _, stride, name = k.split("/", 2)
                rows.append(("D step", name, relerr(au[name].grad.double().reshape(-1)[::int(stride)], gs[k])))
        all_rows("D step", "d_all/", au)
        print("%s (%s oracle on the CPU vs the reference's fp64 gradient samples):" % (tag, seed_note))
        for what, name, e in rows:
            print("   %-7s %-62s %.2e" % (what, name, e))
        for what in ("G step", "D step"):
            es = sorted((e, name) for w_, name, e in allrows if w_ == what)
            print("   %s, all %d parameter gradients (thin samples, floored): median %.1e, worst five: %s"
                  % (what, len(es), es[len(es) // 2][0], ", ".join("%s %.1e" % (nm, e) for e, nm in es[-5:])))
