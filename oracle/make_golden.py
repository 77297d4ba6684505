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


def gen_trainer(tag, reg_param, n_steps=3, n_au_steps=1):
    """Trainer protocol (real im_train_step / au_train_step of the reference) on the tiny
    config for n_steps consecutive iterations, fp64."""
    s, c, d, B, m, n, k = 16, 1, 32, 3, 1, 3, 4
    torch.set_default_dtype(torch.float64)
    au, im = make_models(s, c, d, tag + "/", torch.float64)
    with tempfile.TemporaryDirectory() as td:
        tr = GIMImgTrainer(td, m, n, k, au, im, au_lr=2e-3, im_lr=1e-3, env_noise_mapping_lr=1e-4,
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
    np.savez_compressed(os.path.join(OUT, "trainer_%s.npz" % tag), **st)
    with open(os.path.join(OUT, "trainer_%s.json" % tag), "w") as f:
        json.dump({"config": dict(s=s, c=c, d=d, B=B, m=m, n=n, k=k, reg_param=reg_param, n_au_steps=n_au_steps,
                                  au_lr=2e-3, im_lr=1e-3, noise_lr=1e-4, milestones=[2], gamma=0.5), "meta": meta}, f)
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


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["blocks", "keys", "tiny", "trainer", "bench", "gaussian"]
    if "gaussian" in which:
        gen_gaussian()
    if "blocks" in which:
        gen_blocks()
    if "keys" in which:
        gen_keys()
    if "tiny" in which:
        gen_nets("tiny64", 16, 1, 32, 2, 1, 3, 4, torch.float64, full=True)
        gen_nets("tiny_m2", 16, 1, 32, 2, 2, 2, 1, torch.float64, full=False)
        gen_nets("tiny_att", 16, 1, 32, 2, 1, 3, 4, torch.float64, full=False, use_img_att=True)
    if "trainer" in which:
        gen_trainer("reg0", 0.0)
        gen_trainer("reg10", 10.0)
        gen_trainer("nau2", 0.0, n_steps=2, n_au_steps=2)
    if "bench" in which:
        gen_nets("om32_f64", 32, 1, 512, 2, 1, 5, 10, torch.float64, full=False)
        gen_nets("om32_f32", 32, 1, 512, 2, 1, 5, 10, torch.float32, full=False)
        gen_nets("vox64_f64", 64, 3, 512, 1, 1, 5, 10, torch.float64, full=False)
        gen_nets("vox64_f32", 64, 3, 512, 1, 1, 5, 10, torch.float32, full=False)
    if "bench128" in which:   # BASELINE config 5 shape (m > 1 leaked images), fp64
        gen_nets("vox128_f64", 128, 3, 512, 1, 2, 2, 3, torch.float64, full=False)


if __name__ == "__main__":
    main()
