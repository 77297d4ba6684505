"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (``/root/reference`` present); the GPU box never
sees the reference.  Harness per SURVEY.md section 8(c): stub the missing
third-party modules (colorama, torchvision, tensorboardX), never write bytecode
into the reference tree, no-op torch's InstanceNorm spatial-size check (torch
1.2.0 had none), inject the latent noise ``z`` by temporarily
replacing ``torch.randn``.

Fixtures hold OUTPUTS only: all weights and inputs are re-creatable from
``oracle/portable_fill.py`` by name.

    python oracle/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import portable_fill as pf  # noqa: E402


def _install_stubs():
    col = types.ModuleType("colorama")
    col.Fore = types.SimpleNamespace(YELLOW="", RESET="")
    sys.modules["colorama"] = col
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvu = types.ModuleType("torchvision.utils")
    tv.transforms, tv.utils = tvt, tvu
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.utils": tvu})
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["tensorboardX"] = tbx
    torch.nn.functional._verify_spatial_size = lambda size: None


_install_stubs()
sys.path.insert(0, REF)
import models.model_blocks as mb  # noqa: E402
import models.gim_basic_models as gbm  # noqa: E402
import models.gim_img_models as gim  # noqa: E402
from training.gim_img_trainer import GIMImgTrainer  # noqa: E402
from training.utils import DataParallelMock  # noqa: E402
import training.gim_img_training as git_  # noqa: E402


class inject_randn:
    """Make the next torch.randn calls return the given tensor (z injection)."""

    def __init__(self, z):
        self.z = z

    def __enter__(self):
        self._orig = torch.randn
        z = self.z

        def fake(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            assert shape == tuple(z.shape), (shape, z.shape)
            return z.clone()
        torch.randn = fake

    def __exit__(self, *a):
        torch.randn = self._orig


def T(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def fill_module(mod, tag, dtype=torch.float64):
    sd = mod.state_dict()
    filled = pf.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], tag)
    mod.to(dtype)  # before loading: load_state_dict copies INTO the module's dtype
    mod.load_state_dict({k: T(v, dtype) for k, v in filled.items()})
    return mod


def grads_of(mod):
    return {k: p.grad.detach().numpy().copy() for k, p in mod.named_parameters() if p.grad is not None}


def buffers_of(mod):
    return {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()
            if k.endswith("weight_u") or k.endswith("weight_v")}


def run_block(store, name, mod, inputs, call, training=True):
    """inputs: {arg: shape}; values from portable normal keyed '<name>/<arg>'."""
    fill_module(mod, name + "/")
    mod.train(training)
    xs = {k: T(pf.normal("%s/%s" % (name, k), s)).requires_grad_() for k, s in inputs.items()}
    y = call(mod, **xs)
    r = T(pf.uniform(name + "/r", tuple(y.shape)))
    (y * r).sum().backward()
    store[name + "/y"] = y.detach().numpy()
    for k, x in xs.items():
        store["%s/d_%s" % (name, k)] = x.grad.numpy()
    for k, g in grads_of(mod).items():
        store["%s/g/%s" % (name, k)] = g
    for k, b in buffers_of(mod).items():
        store["%s/b/%s" % (name, k)] = b


def gen_blocks():
    st = {}
    run_block(st, "resdown3", mb.ResBlockDown(4, 8), {"x": (2, 4, 8, 8)}, lambda m, x: m(x))
    run_block(st, "resdown9", mb.ResBlockDown(3, 8, conv_size=9, padding_size=4), {"x": (2, 3, 16, 16)}, lambda m, x: m(x))
    run_block(st, "resup", mb.ResBlockUp(8, 4), {"x": (2, 8, 4, 4)}, lambda m, x: m(x))
    run_block(st, "resup1x1", mb.ResBlockUp(8, 4), {"x": (3, 8, 1, 1)}, lambda m, x: m(x))
    run_block(st, "adares", mb.AdaResBlock2(8, 6), {"x": (2, 8, 4, 4), "style": (2, 6)}, lambda m, x, style: m(x=x, style=style))
    run_block(st, "adaresup3", mb.AdaResBlockUp2(8, 4, 6), {"x": (2, 8, 4, 4), "style": (2, 6)}, lambda m, x, style: m(x=x, style=style))
    run_block(st, "adaresup9", mb.AdaResBlockUp2(8, 3, 6, conv_size=9, padding_size=4), {"x": (2, 8, 8, 8), "style": (2, 6)},
              lambda m, x, style: m(x=x, style=style))
    run_block(st, "selfatt", mb.SelfAttention(16), {"x": (2, 16, 4, 4)}, lambda m, x: m(x))
    run_block(st, "selfatt_eval", mb.SelfAttention(16), {"x": (2, 16, 4, 4)}, lambda m, x: m(x), training=False)
    run_block(st, "mlp", mb.MLP((6, 10, 12, 4)), {"x": (5, 6)}, lambda m, x: m(x))
    run_block(st, "imgatt", mb.ImgAttention(3, 3), {"x1": (2, 3, 16, 16), "x2": (2, 3, 16, 16)}, lambda m, x1, x2: m(x1, x2))
    stat = gbm.GIMMeanStdFcStat(style_dim=8, fc_n_stats=2, fc_hidden_layers=(16, 24, 16))
    run_block(st, "stat", stat, {"x": (3, 5, 8)}, lambda m, x: m(x))
    run_block(st, "stat_k1", gbm.GIMMeanStdFcStat(8, 2, (16, 24, 16)), {"x": (3, 1, 8)}, lambda m, x: m(x))
    dis = gim.GIMFaceDis(8, 8, gbm.GIMMeanStdFcStat(8, 2, (16, 24, 16)))
    run_block(st, "dis", dis, {"test_src": (3, 5, 8), "test_env": (3, 5, 8), "si_src": (3, 4, 8), "si_env": (3, 4, 8)},
              lambda m, **kw: m(**kw))
    # functional ada_in and custom_std
    x = T(pf.normal("ada_in/x", (2, 5, 4, 4))).requires_grad_()
    ms = T(pf.normal("ada_in/mean", (2, 5, 1))).requires_grad_()
    ss = T(pf.normal("ada_in/std", (2, 5, 1))).requires_grad_()
    y = mb.ada_in(x, ms, ss)
    (y * T(pf.uniform("ada_in/r", tuple(y.shape)))).sum().backward()
    st.update({"ada_in/y": y.detach().numpy(), "ada_in/d_x": x.grad.numpy(), "ada_in/d_mean": ms.grad.numpy(),
               "ada_in/d_std": ss.grad.numpy()})
    x = T(pf.normal("custom_std/x", (3, 5, 7)))
    st["custom_std/y"] = mb.custom_std(x).numpy()
    st["custom_std/y1"] = mb.custom_std(x[:, :1]).numpy()
    # spectral-norm sequence: 3 training calls then 1 eval call on one conv
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(4, 6, 3, padding=1))
    fill_module(conv, "snseq/")
    xin = T(pf.normal("snseq/x", (2, 4, 5, 5)))
    for i in range(4):
        conv.train(i < 3)
        y = conv(xin)
        st["snseq/y%d" % i] = y.detach().numpy()
        st["snseq/w%d" % i] = conv.weight.detach().numpy()
        st["snseq/u%d" % i] = conv.weight_u.detach().numpy().copy()
        st["snseq/v%d" % i] = conv.weight_v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **st)
    print("blocks.npz:", len(st), "arrays")


def key_list(mod):
    return [[k, list(v.shape)] for k, v in mod.state_dict().items()]


def param_names(mod):
    return [k for k, _ in mod.named_parameters()]


def gen_keys():
    out = {}
    for (s, c, d) in [(16, 1, 32), (32, 1, 512), (64, 3, 512), (128, 3, 512)]:
        au = gim.get_au(s, c, d)
        im = gim.get_im(s, c, d)
        out["%d_%d_%d" % (s, c, d)] = {
            "au": key_list(au), "im": key_list(im),
            "au_params": param_names(au), "im_params": param_names(im),
            "im_groups": [len(list(getattr(im, g).parameters())) for g in
                          ("src_encoder", "env_encoder", "env_decoder", "img2img", "img_att", "env_noise_mapper")],
        }
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("state_dict_keys.json written")


def make_models(s, c, d, tag, dtype, use_img_att=False):
    au = fill_module(gim.get_au(s, c, d), tag + "au/", dtype)
    im = fill_module(gim.get_im(s, c, d, use_img_att=use_img_att), tag + "im/", dtype)
    return au, im


def episode(tag, B, m, n, k, c, s, d, dtype):
    def img(name, t):
        return T(np.clip(pf.normal("%s/%s" % (tag, name), (B, t, c, s, s)) * 0.5, -1, 1), dtype)
    return img("leaked", m), img("real", n), img("si", k), T(pf.normal(tag + "/z", (B, n, d)), dtype)


def grad_norms(mod):
    return {k: float(p.grad.double().norm()) for k, p in mod.named_parameters() if p.grad is not None}


def gen_nets(tag, s, c, d, B, m, n, k, dtype, full, use_img_att=False):
    """Whole-net forward/backward of the two agents (impersonator_forward then
    authenticator_forward, no optimiser) with outputs and per-tensor grad norms."""
    torch.set_default_dtype(dtype)
    st, meta = {}, {}
    au, im = make_models(s, c, d, tag + "/", dtype, use_img_att)
    leaked, real, si, z = episode(tag, B, m, n, k, c, s, d, dtype)
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
    au.train(); im.train()
    with inject_randn(z):
        loss, fake, out = tr.forward(mode="impersonator_forward", leaked_sample=leaked, si_sample=si)
    loss.mean().backward()
    st["g/loss"], st["g/out"] = loss.detach().numpy(), out.detach().numpy()
    st["g/fake"] = fake.detach().numpy() if full else fake.detach().numpy()[:1, :2]
    meta["g/im_grad_norms"] = grad_norms(im)
    meta["g/au_grad_norms"] = grad_norms(au)
    if full:
        for kk, g in grads_of(im).items():
            if kk.startswith("env_noise_mapper") or "att.gamma" in kk or kk.endswith("conv_r1.weight_orig"):
                st["g/grad/" + kk] = g
    au.zero_grad(); im.zero_grad()
    res = tr.forward(mode="authenticator_forward", fake_sample=fake.detach(), real_sample=real, si_sample=si)
    res[0].mean().backward()
    for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake", "pred_real", "pred_fake"]):
        st["d/" + nm] = res[i].detach().numpy()
    meta["d/au_grad_norms"] = grad_norms(au)
    if full:
        for kk, g in grads_of(au).items():
            if kk.startswith("dis.mlp") or "att.gamma" in kk or kk.endswith("down_blocks.0.conv_r1.weight_orig"):
                st["d/grad/" + kk] = g
    for kk, b in buffers_of(au).items():
        if "down_blocks.0.conv_r1" in kk:
            st["d/buf/" + kk] = b
    np.savez_compressed(os.path.join(OUT, "nets_%s.npz" % tag), **st)
    with open(os.path.join(OUT, "nets_%s.json" % tag), "w") as f:
        json.dump({"config": dict(s=s, c=c, d=d, B=B, m=m, n=n, k=k, dtype=str(dtype)), "meta": meta}, f)
    torch.set_default_dtype(torch.float32)
    print("nets_%s: done" % tag)


def tensor_stats(sd):
    return {k: [float(v.double().sum()), float(v.double().norm())] for k, v in sd.items()}


def state_samples(prefix, sd, st, cap=512):
    """Strided sample (prime stride, at most `cap` values, float64) of every tensor of a state dict: pins element positions,
    which the per-tensor (sum, norm) pair cannot."""
    for kk, v in sd.items():
        a = v.detach().double().reshape(-1).numpy()
        stride = next(p_ for p_ in (1, 3, 7, 13, 31, 61, 127, 251, 509, 1021, 2039, 4093, 8191, 16381) if p_ * cap >= a.size)
        st["%s/%d/%s" % (prefix, stride, kk)] = a[::stride].copy()


def gen_trainer(tag, reg_param, n_steps=3, n_au_steps=1, lrs=(2e-3, 1e-3, 1e-4)):
    """Trainer protocol (real im_train_step / au_train_step of the reference) on the tiny
    config for n_steps consecutive iterations, fp64.  lrs = (au_lr, im_lr, noise-mapper lr): the R1 fixture runs at the
    REAL learning rates of the path (1e-4 / 1e-4 / 1e-6, train_gim_on_imgs.py defaults): at 2e-3 the reference's own fp32 run leaves
    its fp64 run by 1e-2 on the third iteration (profiles/r02_trainer_fixture_fp32_noise.txt) and the fixture pins nothing there."""
    s, c, d, B, m, n, k = 16, 1, 32, 3, 1, 3, 4
    torch.set_default_dtype(torch.float64)
    au, im = make_models(s, c, d, tag + "/", torch.float64)
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, au_lr=lrs[0], im_lr=lrs[1], env_noise_mapping_lr=lrs[2],
                           lr_milestones=(2,), lr_gamma=0.5, reg_param=reg_param)
    trainer = DataParallelMock(tr)
    st, meta = {}, {"lrs": []}
    for it in range(n_steps):
        leaked, real, si, z = episode("%s/it%d" % (tag, it), B, m, n, k, c, s, d, torch.float64)
        tr.do_global_step()
        tr.update_learning_rate()
        meta["lrs"].append([tr.authenticator_opt.param_groups[0]["lr"], tr.impersonator_opt.param_groups[0]["lr"],
                            tr.impersonator_opt.param_groups[-1]["lr"], tr.global_step])
        with inject_randn(z):
            if (tr.global_step + 1) % n_au_steps == 0:
                g = git_.im_train_step(trainer, leaked, si)
            else:
                g = git_.im_eval_step(trainer, leaked, si)
        dres = git_.au_train_step(trainer, real, g[1], si)
        st["it%d/g_loss" % it], st["it%d/g_out" % it] = g[0].numpy(), g[2].numpy()
        st["it%d/fake" % it] = g[1].numpy()
        for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake", "pred_real", "pred_fake"]):
            st["it%d/d_%s" % (it, nm)] = np.asarray(dres[i].numpy())
    # eval pass afterwards (both agents eval(), no power iteration)
    leaked, real, si, z = episode(tag + "/eval", B, m, n, k, c, s, d, torch.float64)
    with inject_randn(z):
        g = git_.im_eval_step(trainer, leaked, si)
    e = git_.au_eval_step(trainer, real, g[1], si)
    st["eval/g_loss"], st["eval/g_out"], st["eval/d_loss"] = g[0].numpy(), g[2].numpy(), e[0].numpy()
    st["eval/d_out_real"], st["eval/d_out_fake"] = e[4].numpy(), e[5].numpy()
    meta["au_final"] = tensor_stats(au.state_dict())
    meta["im_final"] = tensor_stats(im.state_dict())
    osd = tr.authenticator_opt.state_dict()
    meta["au_opt_steps"] = sorted({int(v["step"]) for v in osd["state"].values()})
    meta["au_opt_n_state"] = len(osd["state"])
    meta["im_opt_n_state"] = len(tr.impersonator_opt.state_dict()["state"])
    meta["im_opt_n_groups"] = len(tr.impersonator_opt.param_groups)
    first = next(iter(tr.authenticator_opt.state.values()))
    meta["au_opt_first_v_norm"] = float(first["exp_avg_sq"].norm())
    # element samples of the state the protocol leaves behind: every parameter / buffer of both agents, and Adam's second
    # moment of every parameter that has one (keyed by parameter NAME)
    state_samples("final/au", au.state_dict(), st)
    state_samples("final/im", im.state_dict(), st)
    for nm, mod, opt in (("au", au, tr.authenticator_opt), ("im", im, tr.impersonator_opt)):
        state_samples("adam_v/" + nm, {kk: opt.state[p_]["exp_avg_sq"] for kk, p_ in mod.named_parameters() if p_ in opt.state}, st)
    np.savez_compressed(os.path.join(OUT, "trainer_%s.npz" % tag), **st)
    with open(os.path.join(OUT, "trainer_%s.json" % tag), "w") as f:
        json.dump({"config": dict(s=s, c=c, d=d, B=B, m=m, n=n, k=k, reg_param=reg_param, n_au_steps=n_au_steps,
                                  au_lr=lrs[0], im_lr=lrs[1], noise_lr=lrs[2], milestones=[2], gamma=0.5), "meta": meta}, f)
    torch.set_default_dtype(torch.float32)
    print("trainer_%s: done" % tag)


def gen_gaussian():
    """BASELINE config 1: the Gaussian toy game (d=10, m=1, n=5, k=10), 5 consecutive iterations of the reference's
    own im_train_step / au_train_step on fixed batches, fp64."""
    import models.gim_gaussian_models as ggm
    from training.gim_gaussian_trainer import GIMGaussianTrainer
    import training.gim_gaussian_training as ggt
    d, B, m, n, k = 10, 64, 1, 5, 10
    torch.set_default_dtype(torch.float64)
    st, keys = {}, {}
    for tag, reg in (("gauss", 0.0), ("gauss_r1", 1.0)):
        au = fill_module(ggm.get_au(d), tag + "/au/")
        im = fill_module(ggm.get_im(d), tag + "/im/")
        keys = {"au": key_list(au), "im": key_list(im), "au_params": param_names(au), "im_params": param_names(im)}
        with tempfile.TemporaryDirectory() as td:
            tr = GIMGaussianTrainer(td, m, n, k, au, im, au_lr=1e-3, im_lr=2e-3, reg_param=reg)
        trainer = DataParallelMock(tr)
        for it in range(5):
            mu = pf.normal("%s/it%d/mu" % (tag, it), (B, 1, d))
            smp = lambda nm, t: T(mu + 0.7 * pf.normal("%s/it%d/%s" % (tag, it, nm), (B, t, d)))  # noqa: E731
            leaked, real, si = smp("leaked", m), smp("real", n), smp("si", k)
            z = T(pf.normal("%s/it%d/z" % (tag, it), (B, n, d)))
            tr.do_global_step()
            with inject_randn(z):
                g = ggt.im_train_step(trainer, leaked, si)
            dres = ggt.au_train_step(trainer, real, g[1], si)
            st["%s/it%d/g_loss" % (tag, it)] = g[0].numpy()
            st["%s/it%d/fake" % (tag, it)] = g[1].numpy()
            st["%s/it%d/g_out" % (tag, it)] = g[2].numpy()
            for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake"]):
                st["%s/it%d/d_%s" % (tag, it, nm)] = np.asarray(dres[i].numpy())
        for kk, v in au.state_dict().items():
            st["%s/final/au/%s" % (tag, kk)] = v.numpy()
        for kk, v in im.state_dict().items():
            st["%s/final/im/%s" % (tag, kk)] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "gaussian.npz"), **st)
    with open(os.path.join(OUT, "gaussian.json"), "w") as f:
        json.dump({"config": dict(d=d, B=B, m=m, n=n, k=k, au_lr=1e-3, im_lr=2e-3, sigma=0.7), "keys": keys}, f)
    torch.set_default_dtype(torch.float32)
    print("gaussian: done")


def gen_gaussian_loop():
    """BASELINE config 1's CALLER: the reference's own train() (training/gim_gaussian_training.py:50-151) for 5 iterations on the
    CPU with a seeded default generator - it draws mu and the three sample sets per iteration with torch.normal and the latent
    z with torch.randn inside the impersonator (models/gim_gaussian_models.py:81), all from that one generator - recording every
    logger call (float32, the dtype the reference runs in); plus known answers of theory/theoretic_game_value.py:10-20 (game_value_mnk), the quantity the toy game converges to."""
    import models.gim_gaussian_models as ggm
    from training.gim_gaussian_trainer import GIMGaussianTrainer
    import training.gim_gaussian_training as ggt
    from theory.theoretic_game_value import game_value_mnk
    d, B, m, n, k = 10, 64, 1, 5, 10
    cfg = dict(d=d, B=B, m=m, n=n, k=k, au_lr=1e-3, im_lr=2e-3, src_sigma=1.0, prior_sigma=2.0, n_iters=5, seed=77,
               save_stats_every=2, save_every=4, reg_param=0.0)
    tag = "gauss_loop"
    # float32, the reference's native dtype: torch.normal draws a different stream in float64, and the draws are part of the pin
    torch.set_default_dtype(torch.float32)
    au = fill_module(ggm.get_au(d), tag + "/au/", torch.float32)
    im = fill_module(ggm.get_im(d), tag + "/im/", torch.float32)
    rec = _Recorder()
    saves = []
    st = {}
    with tempfile.TemporaryDirectory() as td:
        tr = GIMGaussianTrainer(td, m, n, k, au, im, au_lr=cfg["au_lr"], im_lr=cfg["im_lr"], reg_param=cfg["reg_param"])
        tr.save = lambda: saves.append(int(tr.global_step))
        trainer = DataParallelMock(tr)
        torch.manual_seed(cfg["seed"])
        ggt.tqdm = lambda it, **kw: it
        ggt.train(device="cpu", trainer=trainer, logger=rec, n_iters=cfg["n_iters"], batch_size=B, src_dim=d, src_sigma=cfg["src_sigma"],
                  prior_sigma=cfg["prior_sigma"], save_stats_every=cfg["save_stats_every"], save_every=cfg["save_every"])
    for kk, v in au.state_dict().items():
        st["final/au/" + kk] = v.numpy()
    for kk, v in im.state_dict().items():
        st["final/im/" + kk] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "gaussian_loop.npz"), **st)
    kats = [[m_, n_, d_, k_, float(game_value_mnk(m_, n_, d_, k_))] for (m_, n_, d_, k_) in
            ((1, 5, 10, 10), (1, 5, 100, 10), (1, 1, 10, 10), (5, 2, 10, 3), (2, 10, 4, 1), (1, 100, 1, 1000), (3, 4, 64, 7))]
    with open(os.path.join(OUT, "gaussian_loop.json"), "w") as f:
        json.dump({"config": cfg, "scalars": rec.scalars, "saves": saves, "final_global_step": int(tr.global_step),
                   "game_value_mnk": kats}, f)
    torch.set_default_dtype(torch.float32)
    print("gaussian_loop: %d scalars, saves %s" % (len(rec.scalars), saves))


def gen_subnets():
    """the three sub-networks of the generator / authenticator ON THEIR OWN (tiny config 16x16x1, style 32,
    conditioned fill), so that a failure localises below "G step": outputs, input gradients, every parameter gradient and the
    spectral-norm buffers after the call.  models/gim_img_models.py:19-57 (Encoder), :63-95 (EnvDecoder), :218-257 (AdaInImage2Image)."""
    st = {}
    run_block(st, "encoder", gim.Encoder(16, 1, 32), {"x": (3, 1, 16, 16)}, lambda m, x: m(x))
    run_block(st, "envdecoder", gim.EnvDecoder(16, 1, 32), {"x": (3, 32)}, lambda m, x: m(x))
    run_block(st, "img2img", gim.AdaInImage2Image(img_size=16, in_channels=2, out_channels=1, style_dim=32),
              {"x": (3, 2, 16, 16), "style": (3, 32)}, lambda m, x, style: m(x=x, style=style))
    np.savez_compressed(os.path.join(OUT, "subnets.npz"), **st)
    print("subnets.npz:", len(st), "arrays")


# full gradient tensors at the benchmark shapes would be tens of MB per fixture; a fixed strided subsample of the flattened
# [Cout, Cin, kh, kw] gradient pins the position of every kept element (a per-tensor norm is blind to tap / channel order).
# (the numbers in the table are only a note of each tensor's role; the stride is chosen from the tensor's size, see sample())
BENCH_GRAD_SAMPLES = {
    "im": {"img2img.down_block.down_blocks.0.conv_r2.weight_orig": 7,          # the 9x9 64 -> 64 (32x32x1: 9x9 64 -> 64 too)
           "img2img.adain_res_block.res_blocks.2.conv1.weight_orig": 61,       # a 512 -> 512 3x3
           "img2img.adain_up_block.up_blocks.0.conv_l1.weight_orig": 3,        # a 1x1 skip conv
           "env_decoder.up_blocks.2.conv_r1.weight_orig": 31,                  # sub-pixel (upsampled) 3x3
           "img2img.adain_res_block.res_blocks.0.lin1_std.weight": 5},         # a style linear
    "au": {"src_encoder.down_blocks.%(last)d.conv_r2.weight_orig": 61,         # the 512 -> 512 3x3 (pool-folded)
           "env_encoder.down_blocks.1.conv_l1.weight_orig": 1,                 # a 1x1 skip conv (full)
           "src_encoder.down_blocks.0.conv_r1.weight_orig": 1,                 # the image layer (3 / 1 input channels, full)
           "src_encoder.att.conv_h.weight_orig": 3,
           "dis.stat.fc.stat.model.2.weight": 37},
}


def gen_bench_grads(tag, s, c, d, B, m, n, k, dtype):
    """Same weights, inputs and protocol as gen_nets(tag, ...): strided samples of selected gradient TENSORS of the G step and
    of the D step at a benchmark shape (the fixture nets_<tag>.npz pins losses / logits / per-tensor gradient norms)."""
    torch.set_default_dtype(dtype)
    st = {}
    au, im = make_models(s, c, d, tag + "/", dtype)
    leaked, real, si, z = episode(tag, B, m, n, k, c, s, d, dtype)
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
    au.train(); im.train()
    with inject_randn(z):
        loss, fake, out = tr.forward(mode="impersonator_forward", leaked_sample=leaked, si_sample=si)
    loss.mean().backward()
    last = len(au.src_encoder.down_blocks) - 1
    gi = grads_of(im)

    def sample(prefix, name, arr, cap=20000):   # at most ~20 000 values per tensor, prime stride, float32 (compared at 1e-3)
        stride = next(p_ for p_ in (1, 3, 7, 13, 31, 61, 127, 251, 509, 1021, 2039, 4093, 8191) if p_ * cap >= arr.size)
        st["%s/%d/%s" % (prefix, stride, name)] = arr.reshape(-1)[::stride].astype(np.float32)
    for name in BENCH_GRAD_SAMPLES["im"]:
        sample("g", name, gi[name])
    for name, arr in gi.items():     # EVERY parameter gradient of the G step, thinner (<= 1500 values each)
        sample("g_all", name, arr, cap=1500)
    au.zero_grad(); im.zero_grad()
    res = tr.forward(mode="authenticator_forward", fake_sample=fake.detach(), real_sample=real, si_sample=si)
    res[0].mean().backward()
    ga = grads_of(au)
    for name in BENCH_GRAD_SAMPLES["au"]:
        name = name % {"last": last}
        sample("d", name, ga[name])
    for name, arr in ga.items():     # EVERY parameter gradient of the D step
        sample("d_all", name, arr, cap=1500)
    np.savez_compressed(os.path.join(OUT, "nets_%s_grads.npz" % tag), **st)
    torch.set_default_dtype(torch.float32)
    print("nets_%s_grads: %d arrays, %d values" % (tag, len(st), sum(v.size for v in st.values())))


def gen_ckpt():
    """A checkpoint WRITTEN BY THE REFERENCE (training/checkpoints.py:21-44 through GIMImgTrainer.save, training/gim_img_trainer.py:
    163-172): model + torch.optim.Adam state + GlobalStep after 3 iterations of the reference's own loop on a small config
    (16x16x1, style 16, fp32 - the reference's native dtype; ~0.8 MB), and what the reference's 4th iteration then returns.
    The product must load the file with resume_from_ckpt and reproduce that 4th iteration."""
    import shutil
    s, c, d, B, m, n, k = 16, 1, 16, 3, 1, 3, 4
    tag = "ckpt"
    torch.set_default_dtype(torch.float32)
    au, im = make_models(s, c, d, tag + "/", torch.float32)
    st = {}
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, au_lr=1e-4, im_lr=1e-4, env_noise_mapping_lr=1e-6, reg_param=0.0)
        trainer = DataParallelMock(tr)
        for it in range(4):
            leaked, real, si, z = episode("%s/it%d" % (tag, it), B, m, n, k, c, s, d, torch.float32)
            tr.do_global_step()
            tr.update_learning_rate()
            with inject_randn(z):
                g = git_.im_train_step(trainer, leaked, si)
            dres = git_.au_train_step(trainer, real, g[1], si)
            if it == 2:
                tr.save(epoch=0)
                src = os.path.join(td, "ckpts", "model_%08d.pt" % tr.global_step)
                assert tr.global_step == 2 and os.path.exists(src)
                shutil.copy(src, os.path.join(OUT, "ref_ckpt_model_00000002.pt"))
            if it == 3:
                st["g_loss"], st["g_out"], st["fake"] = g[0].numpy(), g[2].numpy(), g[1].numpy()
                for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake"]):
                    st["d_" + nm] = np.asarray(dres[i].numpy())
    np.savez_compressed(os.path.join(OUT, "ref_ckpt_step4.npz"), **st)
    with open(os.path.join(OUT, "ref_ckpt.json"), "w") as f:
        json.dump({"config": dict(s=s, c=c, d=d, B=B, m=m, n=n, k=k, au_lr=1e-4, im_lr=1e-4, noise_lr=1e-6, tag=tag),
                   "au_keys": key_list(au), "im_keys": key_list(im),
                   "size_bytes": os.path.getsize(os.path.join(OUT, "ref_ckpt_model_00000002.pt"))}, f)
    print("ref_ckpt: done")


class _Recorder:
    """Stands in for training/logger.py:12-92 (tensorboardX / torchvision are absent): records what the loop logs."""

    def __init__(self):
        self.scalars, self.imgs = [], []

    def add_scalar(self, category, k, v, global_step):
        self.scalars.append([category, k, int(global_step), float(v)])

    def add_imgs(self, imgs, category, k, global_step, nrow=5):
        self.imgs.append([category, k, int(global_step), list(imgs.shape), float(imgs.double().sum()), float(imgs.double().abs().max())])


class _MemDS(torch.utils.data.Dataset):
    """In-memory episodes with the sample contract of ImgGIMDataSet.__getitem__ (data_handling/img_datasets.py:68-103)."""

    def __init__(self, tag, n_ex, m, n, k, c, s, dtype):
        def img(name, i, t):
            return T(np.clip(pf.normal("%s/%d/%s" % (tag, i, name), (t, c, s, s)) * 0.5, -1, 1), dtype)
        self.ex = [{"real_sample": img("real", i, n), "leaked_sample": img("leaked", i, m), "si_sample": img("si", i, k),
                    "class": i, "class_name": "c%d" % i} for i in range(n_ex)]

    def __len__(self):
        return len(self.ex)

    def __getitem__(self, i):
        return self.ex[i]


class inject_randn_seq:
    """Every torch.randn call inside returns the portable normal keyed by the call's index and shape (latent noise z is drawn
    inside GIMFaceImpersonator.forward, models/gim_img_models.py:374, also by the image dumps of the loop)."""

    def __init__(self, tag, dtype):
        self.tag, self.dtype, self.n = tag, dtype, 0

    def __enter__(self):
        self._orig = torch.randn
        outer = self

        def fake(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            z = T(pf.normal("%s/z%d" % (outer.tag, outer.n), shape), outer.dtype)
            outer.n += 1
            return z
        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self._orig


def gen_loop():
    """The reference's caller loop itself (training/gim_img_training.py:186-354 train_epoch, :98-154 eval_step, :23-73 image dumps)
    for two epochs of two iterations on a fixed in-memory dataset, every cadence set so that it fires: the stream of logged
    scalars, the image-dump events and the checkpoint calls.  fp64, tiny config, n_au_steps = 2 (eval-mode generator passes)."""
    s, c, d, m, n, k = 16, 1, 32, 1, 3, 4
    tag = "loop"
    torch.set_default_dtype(torch.float64)
    au, im = make_models(s, c, d, tag + "/", torch.float64)
    train_ds = _MemDS(tag + "/train", 7, m, n, k, c, s, torch.float64)
    val_ds = _MemDS(tag + "/val", 4, m, n, k, c, s, torch.float64)
    rec = _Recorder()
    saves = []
    cfg = dict(train_batch_size=3, val_batch_size=2, save_every=3, eval_every=2, save_imgs_every=2, train_eval_indices=[0, 5],
               val_eval_indices=[1], tb_log_every=1, tb_log_enc_every=2, n_au_steps=2, n_epochs=2, seed=123,
               au_lr=2e-3, im_lr=1e-3, noise_lr=1e-4, milestones=[2], gamma=0.5)
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, au_lr=cfg["au_lr"], im_lr=cfg["im_lr"], env_noise_mapping_lr=cfg["noise_lr"],
                           lr_milestones=tuple(cfg["milestones"]), lr_gamma=cfg["gamma"], reg_param=0.0)
        orig_save = tr.save
        tr.save = lambda epoch: (saves.append([int(tr.global_step), int(epoch)]), orig_save(epoch=epoch))[1]
        trainer = DataParallelMock(tr)
        torch.manual_seed(cfg["seed"])     # the DataLoader's shuffle draws its seed from the default generator
        with inject_randn_seq(tag, torch.float64) as inj:
            for ep in range(cfg["n_epochs"]):
                git_.train_epoch(device="cpu", logger=rec, epoch=ep, trainer=trainer, train_ds=train_ds, val_ds=val_ds,
                                 train_batch_size=cfg["train_batch_size"], val_batch_size=cfg["val_batch_size"], num_workers=0,
                                 save_every=cfg["save_every"], eval_every=cfg["eval_every"], save_imgs_every=cfg["save_imgs_every"],
                                 train_eval_indices=cfg["train_eval_indices"], val_eval_indices=cfg["val_eval_indices"],
                                 tb_log_every=cfg["tb_log_every"], tb_log_enc_every=cfg["tb_log_enc_every"], n_au_steps=cfg["n_au_steps"])
            n_z = inj.n
        ckpts = sorted(os.listdir(os.path.join(td, "ckpts")))
    with open(os.path.join(OUT, "loop.json"), "w") as f:
        json.dump({"config": dict(cfg, s=s, c=c, d=d, m=m, n=n, k=k, n_train=7, n_val=4), "scalars": rec.scalars, "imgs": rec.imgs,
                   "saves": saves, "ckpt_files": ckpts, "n_randn_calls": n_z, "final_global_step": int(tr.global_step),
                   "au_final": tensor_stats(au.state_dict()), "im_final": tensor_stats(im.state_dict())}, f)
    torch.set_default_dtype(torch.float32)
    print("loop.json: %d scalars, %d image dumps, saves %s, %d randn calls" % (len(rec.scalars), len(rec.imgs), saves, n_z))


def gen_data():
    """The dataset sample contract (data_handling/img_datasets.py:24-110 ImgGIMDataSet, :270-303 load_image / process_pil_image /
    adjust_dynamic_range): the reference's dataset class reads PNG files written here from portable-fill uint8 images and returns
    its example dicts; the fixture holds the uint8 bank, and per example the source image index and flip flag of every returned
    image (recovered by exact match) next to the returned float tensors.
    torchvision==0.4.0 (requirements.txt:14) is absent from the image: the three transforms the path touches are supplied per
    their published behaviour - ToTensor (uint8 HWC -> float32 CHW / 255), RandomHorizontalFlip (python random() < 0.5 ->
    PIL FLIP_LEFT_RIGHT), Compose - everything else executed is the reference's own code."""
    import random
    from PIL import Image
    tvt = sys.modules["torchvision.transforms"]

    class ToTensor:
        def __call__(self, pic):
            a = np.array(pic, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255)

    class RandomHorizontalFlip:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, img):
            return img.transpose(Image.FLIP_LEFT_RIGHT) if random.random() < self.p else img

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, img):
            for t in self.ts:
                img = t(img)
            return img
    tvt.ToTensor, tvt.RandomHorizontalFlip, tvt.Compose = ToTensor, RandomHorizontalFlip, Compose
    import data_handling.img_datasets as ids
    S, C, m, n, k = 8, 3, 1, 3, 4
    sizes = [9, 8, 11, 5]            # the last class has fewer than m + n + k images: filtered out (img_datasets.py:59-61)
    bank, offs = [], [0]
    with tempfile.TemporaryDirectory() as root:
        for ci, sz in enumerate(sizes):
            os.makedirs(os.path.join(root, "train", "cls%02d" % ci))
            for j in range(sz):
                a = (pf.uniform("data/%d/%d" % (ci, j), (S, S, C), 0.0, 256.0)).astype(np.uint8)
                Image.fromarray(a, "RGB").save(os.path.join(root, "train", "cls%02d" % ci, "img%03d.png" % j))
                bank.append(a)
            offs.append(offs[-1] + sz)
        bank = np.stack(bank)
        ds = ids.ImgGIMDataSet(root=root, split="train", img_channels=C, img_size=S, m=m, n=n, si=k, example_cnt_per_class=2,
                               img_suffix=".png", mirror=True)
        st = {"bank": bank, "offsets": np.asarray(offs), "len": np.asarray(len(ds)), "n_classes": np.asarray(ds.n_classes)}
        names = sorted(os.listdir(os.path.join(root, "train")))
        random.seed(11)
        meta = []
        for e, index in enumerate([0, 3, 5, 2]):
            ex = ds[index]
            cls_dir = ex["class_name"]
            ci = names.index(cls_dir)
            meta.append({"index": index, "class": int(ex["class"]), "class_name": cls_dir, "bank_class": ci})
            for part in ("leaked_sample", "real_sample", "si_sample"):
                t = ex[part]
                st["ex%d/%s" % (e, part)] = t.numpy()
                src, flips = [], []
                for img in t:     # which bank image, flipped or not: exact match in uint8 space
                    u8 = np.rint((img.numpy().transpose(1, 2, 0) + 1.0) * 127.5).astype(np.uint8)
                    hit = [(j, f) for j in range(offs[ci], offs[ci + 1]) for f in (0, 1)
                           if np.array_equal(bank[j][:, ::-1] if f else bank[j], u8)]
                    assert len(hit) == 1, hit
                    src.append(hit[0][0]); flips.append(hit[0][1])
                st["ex%d/%s/src" % (e, part)] = np.asarray(src, dtype=np.int32)
                st["ex%d/%s/flip" % (e, part)] = np.asarray(flips, dtype=np.uint8)
        # process_pil_image with a real resize (12x12 -> 8x8 bilinear) and adjust_dynamic_range on its own
        big = (pf.uniform("data/big", (12, 12, C), 0.0, 256.0)).astype(np.uint8)
        st["resize/in"] = big
        st["resize/out"] = ids.process_pil_image(Image.fromarray(big, "RGB"), img_size=S).numpy()
        st["adr/out"] = ids.adjust_dynamic_range(torch.from_numpy(bank[:2].astype(np.float32) / np.float32(255.0)), (0., 1.), (-1, 1)).numpy()
    np.savez_compressed(os.path.join(OUT, "data.npz"), **st)
    with open(os.path.join(OUT, "data.json"), "w") as f:
        json.dump({"config": dict(S=S, C=C, m=m, n=n, k=k, sizes=sizes, example_cnt_per_class=2, python_random_seed=11), "examples": meta}, f)
    print("data.npz: %d arrays" % len(st))


def gen_data_omniglot():
    """The Omniglot sample contract (data_handling/img_datasets.py:118-215 OmniglotGIMDataSet): alphabet / character directories,
    every image loaded ONCE at construction in mode 'L' (one channel) without augmentation, 20 images per character, an example =
    m + n + si distinct images of one character drawn with random.sample, ValueError beyond 20.  The reference's class reads PNG
    files written here from portable-fill uint8 images; the fixture holds the uint8 bank in the reference's own class order
    (os.listdir order of the temporary tree, recorded), and per example the bank index of every returned image (exact match)
    next to the returned float tensors.  Same torchvision stand-ins as gen_data (ToTensor only is touched)."""
    import random
    from PIL import Image
    tvt = sys.modules["torchvision.transforms"]
    if not hasattr(tvt, "ToTensor"):
        class ToTensor:
            def __call__(self, pic):
                a = np.array(pic, dtype=np.uint8)
                if a.ndim == 2:
                    a = a[:, :, None]
                return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255)
        tvt.ToTensor = ToTensor
    import data_handling.img_datasets as ids
    S, m, n, k, per = 8, 1, 5, 10, 20
    tree = {"alphaB": ["ch2", "ch1"], "alphaA": ["ch3"]}
    with tempfile.TemporaryDirectory() as root:
        for a_, chars in tree.items():
            for ch in chars:
                os.makedirs(os.path.join(root, "train", a_, ch))
                for j in range(per):
                    img = (pf.uniform("omni/%s/%s/%d" % (a_, ch, j), (S, S), 0.0, 256.0)).astype(np.uint8)
                    Image.fromarray(img, "L").save(os.path.join(root, "train", a_, ch, "img%02d.png" % j))
        ds = ids.OmniglotGIMDataSet(root=root, split="train", img_channels=1, img_size=S, m=m, n=n, si=k, example_cnt_per_class=3)
        # the bank in the reference's class order; within a class in ITS file order (list_files = os.listdir order)
        bank, offs, names = [], [0], []
        for ci, character in enumerate(ds._characters):
            names.append(character)
            for img_name, _ in ds._character_images[ci]:
                bank.append(np.array(Image.open(os.path.join(root, "train", character, img_name)).convert("L"), dtype=np.uint8)[:, :, None])
            offs.append(len(bank))
        bank = np.stack(bank)
        st = {"bank": bank, "offsets": np.asarray(offs), "len": np.asarray(len(ds)), "n_classes": np.asarray(ds.n_classes)}
        random.seed(5)
        meta = []
        for e, index in enumerate([0, 4, 8, 7]):
            ex = ds[index]
            ci = int(ex["class"])
            meta.append({"index": index, "class": ci, "class_name": ex["class_name"]})
            for part in ("leaked_sample", "real_sample", "si_sample"):
                t = ex[part]
                st["ex%d/%s" % (e, part)] = t.numpy()
                src = []
                for img in t:
                    u8 = np.rint((img.numpy().transpose(1, 2, 0) + 1.0) * 127.5).astype(np.uint8)
                    hit = [j for j in range(offs[ci], offs[ci + 1]) if np.array_equal(bank[j], u8)]
                    assert len(hit) == 1, hit
                    src.append(hit[0])
                st["ex%d/%s/src" % (e, part)] = np.asarray(src, dtype=np.int32)
        try:
            ids.OmniglotGIMDataSet(root=root, split="train", img_channels=1, img_size=S, m=1, n=10, si=10, example_cnt_per_class=1)
            too_many = None
        except ValueError as err:
            too_many = str(err)
    np.savez_compressed(os.path.join(OUT, "data_omniglot.npz"), **st)
    with open(os.path.join(OUT, "data_omniglot.json"), "w") as f:
        json.dump({"config": dict(S=S, C=1, m=m, n=n, k=k, per_class=per, example_cnt_per_class=3, python_random_seed=5),
                   "class_names": names, "examples": meta, "too_many_error": too_many}, f)
    print("data_omniglot.npz: %d arrays; classes %s" % (len(st), names))


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["blocks", "keys", "tiny", "trainer", "bench", "gaussian", "gaussloop", "subnets", "benchgrads", "ckpt", "loop", "data",
                             "omniglot"]
    if "gaussian" in which:
        gen_gaussian()
    if "gaussloop" in which:
        gen_gaussian_loop()
    if "blocks" in which:
        gen_blocks()
    if "keys" in which:
        gen_keys()
    if "tiny" in which:
        gen_nets("tiny64", 16, 1, 32, 2, 1, 3, 4, torch.float64, full=True)
        gen_nets("tiny_m2", 16, 1, 32, 2, 2, 2, 1, torch.float64, full=False)
        gen_nets("tiny_att", 16, 1, 32, 2, 1, 3, 4, torch.float64, full=False, use_img_att=True)
    if "trainer" in which:
        # all three at the REAL learning rates of the path (train_gim_on_imgs.py defaults); round 4 moved reg0 / nau2 there too:
        # at 2e-3 / 1e-3 Adam (beta1 = 0) moves every weight by ~lr per update, and the element samples of the second moments
        # sat 14x above the reference's own fp32-vs-fp64 floor (3.3e-3 vs 2.4e-4, profiles/r04_trainer_fixture_fp32_noise.txt)
        gen_trainer("reg0", 0.0, lrs=(1e-4, 1e-4, 1e-6))
        gen_trainer("reg10", 10.0, lrs=(1e-4, 1e-4, 1e-6))
        gen_trainer("nau2", 0.0, n_steps=2, n_au_steps=2, lrs=(1e-4, 1e-4, 1e-6))
    if "bench" in which:
        gen_nets("om32_f64", 32, 1, 512, 2, 1, 5, 10, torch.float64, full=False)
        gen_nets("om32_f32", 32, 1, 512, 2, 1, 5, 10, torch.float32, full=False)
        gen_nets("vox64_f64", 64, 3, 512, 1, 1, 5, 10, torch.float64, full=False)
        gen_nets("vox64_f32", 64, 3, 512, 1, 1, 5, 10, torch.float32, full=False)
    if "bench128" in which:   # BASELINE config 5 shape (m > 1 leaked images), fp64
        gen_nets("vox128_f64", 128, 3, 512, 1, 2, 2, 3, torch.float64, full=False)
    if "cfg5" in which:       # BASELINE config 5 AS STATED: 128x128x3, m=5 n=20 k=20 (one episode; ~13 TFLOP in fp64 on the CPU)
        gen_nets("vox128_m5n20k20", 128, 3, 512, 1, 5, 20, 20, torch.float64, full=False)
    if "subnets" in which:
        gen_subnets()
    if "benchgrads" in which:
        gen_bench_grads("vox64_f64", 64, 3, 512, 1, 1, 5, 10, torch.float64)
        gen_bench_grads("om32_f64", 32, 1, 512, 2, 1, 5, 10, torch.float64)
    if "ckpt" in which:
        gen_ckpt()
    if "loop" in which:
        gen_loop()
    if "data" in which:
        gen_data()
    if "omniglot" in which:
        gen_data_omniglot()


if __name__ == "__main__":
    main()
