"""How far the REFERENCE'S OWN arithmetic in fp32 (the oracle run in float32 on the CPU) drifts from its fp64 run on the trainer
protocol fixtures (tests/golden/trainer_*.npz), per iteration: the noise floor any fp32 implementation sits on.  The R1 fixture
(reg_param = 10, lr 2e-3 / 1e-3, Adam beta1 = 0) is chaotic from iteration 2 on.   python tools/trainer_fixture_fp32_noise.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gim_oracle as go  # noqa: E402
from tests.helpers import episode, filled_sd, load_json, load_keys, load_npz, relerr  # noqa: E402

for tag in ("reg0", "reg10", "nau2"):
    g, meta = load_npz("trainer_%s.npz" % tag), load_json("trainer_%s.json" % tag)
    c = meta["config"]
    keys = load_keys("16_1_32")
    au = {k: v.float() for k, v in filled_sd(keys["au"], tag + "/au/").items()}
    im = {k: v.float() for k, v in filled_sd(keys["im"], tag + "/im/").items()}
    tr = go.OracleTrainer(au, im, c["n"], c["au_lr"], c["im_lr"], c["noise_lr"], reg_param=c["reg_param"])
    rows = []
    for it in range(len(meta["meta"]["lrs"])):
        leaked, real, si, z = [t.float() for t in episode("%s/it%d" % (tag, it), c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"])]
        scale = c["gamma"] if it + 1 >= c["milestones"][0] else 1.0
        if (it + 1) % c["n_au_steps"] == 0:
            tr.im_opt.zero_grad()
            loss, fake, gout = go.impersonator_forward(au, im, leaked, si, c["n"], z, tr.au_training, True)
            loss.mean().backward()
            tr.im_opt.step(lr_scale=scale)
        else:
            with torch.no_grad():
                loss, fake, gout = go.impersonator_forward(au, im, leaked, si, c["n"], z, tr.au_training, False)
        tr.au_training = True
        tr.au_opt.zero_grad()
        res = go.authenticator_forward(au, fake.detach(), real, si, True, c["reg_param"])
        res[0].mean().backward()
        tr.au_opt.step(lr_scale=scale)
        rows.append("it%d: g_loss %.1e g_out %.1e fake %.1e d_loss %.1e"
                    % (it, relerr(loss.mean().double(), g["it%d/g_loss" % it]), relerr(gout.detach().double(), g["it%d/g_out" % it]),
                       relerr(fake.double(), g["it%d/fake" % it]), relerr(res[0].detach().mean().double(), g["it%d/d_loss" % it])))
    leaked, real, si, z = [t.float() for t in episode(tag + "/eval", c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"])]
    with torch.no_grad():
        loss, fake, g_out = go.impersonator_forward(au, im, leaked, si, c["n"], z, True, False)
        res = go.authenticator_forward(au, fake, real, si, False, c["reg_param"], grad=False)
    fin = []
    for nm, sd in (("au", au), ("im", im)):
        worst = (0.0, "")
        for k, (s_ref, n_ref) in meta["meta"][nm + "_final"].items():
            t = sd[k].detach().double()
            dev_ = max(abs(float(t.norm()) - n_ref) / max(n_ref, 1e-30), abs(float(t.sum()) - s_ref) / max(n_ref * t.numel() ** 0.5, 1e-30))
            worst = max(worst, (dev_, k))
        fin.append("%s final state: worst tensor %.1e (%s)" % (nm, worst[0], worst[1]))
    # element samples of the final state / of Adam's second moments (fixture keys final/<agent>/<stride>/<name>, adam_v/...)
    for nm, sd, opt in (("au", au, tr.au_opt), ("im", im, tr.im_opt)):
        lr = c[nm + "_lr"]
        w_rel, w_share, w_v = (0.0, ""), (0.0, ""), (0.0, "")
        # tensors whose whole gradient is mathematically zero (a conv bias in front of a norm layer, the attention f-bias): their
        # Adam updates are a random walk of rounding noise in fp64 and fp32 alike - recognised by their second moment, skipped
        walk = {key.split("/", 3)[3] for key in g.files if key.startswith("adam_v/%s/" % nm) and float(np.sqrt(g[key]).mean()) < 1e-6}
        for key in g.files:
            if key.count("/") >= 3 and key.split("/", 3)[3] in walk:
                continue
            if key.startswith("final/%s/" % nm):
                _, _, stride, name = key.split("/", 3)
                mine = sd[name].detach().double().reshape(-1)[::int(stride)].numpy()
                ref = g[key]
                d_ = np.abs(mine - ref)
                w_rel = max(w_rel, (float(np.linalg.norm(mine - ref) / max(np.linalg.norm(ref), 1e-30)), name))
                w_share = max(w_share, (float((d_ > 0.25 * lr).mean()), name))
            elif key.startswith("adam_v/%s/" % nm):
                _, _, stride, name = key.split("/", 3)
                mine = opt.state[name]["v"].detach().double().reshape(-1)[::int(stride)].numpy()
                ref = g[key]
                w_v = max(w_v, (float(np.linalg.norm(mine - ref) / max(np.linalg.norm(ref), 1e-30)), name))
        fin.append("%s samples: worst L2 %.1e (%s), worst share of elements off by > lr/4 %.1e (%s), worst Adam v L2 %.1e (%s)"
                   % (nm, w_rel[0], w_rel[1], w_share[0], w_share[1], w_v[0], w_v[1]))
    first = next(k for k in au if go.is_param(k))
    fin.append("au Adam v of the first parameter (%s): norm %.1e off" % (first, abs(float(tr.au_opt.state[first]["v"].double().norm()) - meta["meta"]["au_opt_first_v_norm"])
                                                                        / meta["meta"]["au_opt_first_v_norm"]))
    print(tag, " | ".join(rows), "|", " ; ".join(fin), "|", "eval: g_loss %.1e g_out %.1e d_loss %.1e d_out_real %.1e d_out_fake %.1e"
          % (relerr(loss.mean().double(), g["eval/g_loss"]), relerr(g_out.double(), g["eval/g_out"]),
             relerr(res[0].mean().double(), g["eval/d_loss"]), relerr(res[4].double().mean(), g["eval/d_out_real"]),
             relerr(res[5].double().mean(), g["eval/d_out_fake"])))
