"""CPU suite: pin the oracle (oracle/gim_oracle.py) against the golden vectors that
oracle/make_golden.py captured from the reference itself (fp64 unless noted)."""
import numpy as np
import pytest
import torch

from oracle import gim_oracle as go
from oracle import portable_fill as pf
from tests.helpers import T, episode, filled_sd, load_json, load_keys, load_npz, relerr

TOL = 1e-10


def block_sd(g, name):
    ks = [(k[len(name) + 3:], g[k].shape) for k in g.files if k.startswith(name + "/g/") or k.startswith(name + "/b/")]
    sd = filled_sd(ks, name + "/")
    go.set_requires_grad(sd)
    return sd


BLOCKS = {
    "resdown3": (dict(x=(2, 4, 8, 8)), lambda sd, x: go.res_block_down(sd, "", x, True)),
    "resdown9": (dict(x=(2, 3, 16, 16)), lambda sd, x: go.res_block_down(sd, "", x, True)),
    "resup": (dict(x=(2, 8, 4, 4)), lambda sd, x: go.res_block_up(sd, "", x, True)),
    "resup1x1": (dict(x=(3, 8, 1, 1)), lambda sd, x: go.res_block_up(sd, "", x, True)),
    "adares": (dict(x=(2, 8, 4, 4), style=(2, 6)), lambda sd, x, style: go.ada_res_block2(sd, "", x, style, True)),
    "adaresup3": (dict(x=(2, 8, 4, 4), style=(2, 6)), lambda sd, x, style: go.ada_res_block_up2(sd, "", x, style, True)),
    "adaresup9": (dict(x=(2, 8, 8, 8), style=(2, 6)), lambda sd, x, style: go.ada_res_block_up2(sd, "", x, style, True)),
    "selfatt": (dict(x=(2, 16, 4, 4)), lambda sd, x: go.self_attention(sd, "", x, True)),
    "selfatt_eval": (dict(x=(2, 16, 4, 4)), lambda sd, x: go.self_attention(sd, "", x, False)),
    "mlp": (dict(x=(5, 6)), lambda sd, x: go.mlp(sd, "", x)),
    "imgatt": (dict(x1=(2, 3, 16, 16), x2=(2, 3, 16, 16)), lambda sd, x1, x2: go.img_attention(sd, "", x1, x2, True)),
    "stat": (dict(x=(3, 5, 8)), lambda sd, x: go.mean_std_fc_stat(sd, "", x)),
    "stat_k1": (dict(x=(3, 1, 8)), lambda sd, x: go.mean_std_fc_stat(sd, "", x)),
    "dis": (dict(test_src=(3, 5, 8), test_env=(3, 5, 8), si_src=(3, 4, 8), si_env=(3, 4, 8)),
            lambda sd, **kw: go.face_dis(sd, "", **kw)),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_matches_reference(name):
    g = load_npz("blocks.npz")
    inputs, fn = BLOCKS[name]
    sd = block_sd(g, name)
    xs = {k: T(pf.normal("%s/%s" % (name, k), s)).requires_grad_() for k, s in inputs.items()}
    y = fn(sd, **xs)
    assert relerr(y, g[name + "/y"]) < TOL
    (y * T(pf.uniform(name + "/r", tuple(y.shape)))).sum().backward()
    for k, x in xs.items():
        assert relerr(x.grad, g["%s/d_%s" % (name, k)]) < 1e-9, k
    for k in g.files:
        if k.startswith(name + "/g/"):
            assert relerr(sd[k[len(name) + 3:]].grad, g[k], atol=1e-6) < 1e-8, k
        if k.startswith(name + "/b/"):
            assert relerr(sd[k[len(name) + 3:]], g[k]) < TOL, k


def test_ada_in_and_custom_std():
    g = load_npz("blocks.npz")
    x = T(pf.normal("ada_in/x", (2, 5, 4, 4))).requires_grad_()
    ms = T(pf.normal("ada_in/mean", (2, 5, 1))).requires_grad_()
    ss = T(pf.normal("ada_in/std", (2, 5, 1))).requires_grad_()
    y = go.ada_in(x, ms, ss)
    assert relerr(y, g["ada_in/y"]) < TOL
    (y * T(pf.uniform("ada_in/r", tuple(y.shape)))).sum().backward()
    assert relerr(x.grad, g["ada_in/d_x"]) < 1e-9
    assert relerr(ms.grad, g["ada_in/d_mean"]) < 1e-9
    assert relerr(ss.grad, g["ada_in/d_std"]) < 1e-9
    x = T(pf.normal("custom_std/x", (3, 5, 7)))
    assert relerr(go.custom_std(x), g["custom_std/y"]) < TOL
    assert float(go.custom_std(x[:, :1]).abs().max()) == 0.0 and float(np.abs(g["custom_std/y1"]).max()) == 0.0


def test_spectral_norm_sequence():
    """3 training calls (one power iteration each) then an eval call (none)."""
    g = load_npz("blocks.npz")
    sd = filled_sd([("bias", (6,)), ("weight_orig", (6, 4, 3, 3)), ("weight_u", (6,)), ("weight_v", (36,))], "snseq/")
    x = T(pf.normal("snseq/x", (2, 4, 5, 5)))
    for i in range(4):
        w = go.sn_weight(sd, "", training=i < 3)
        assert relerr(w, g["snseq/w%d" % i]) < TOL
        assert relerr(sd["weight_u"], g["snseq/u%d" % i]) < TOL
        assert relerr(sd["weight_v"], g["snseq/v%d" % i]) < TOL
    sd2 = filled_sd([("bias", (6,)), ("weight_orig", (6, 4, 3, 3)), ("weight_u", (6,)), ("weight_v", (36,))], "snseq/")
    assert relerr(go.sn_conv(sd2, "", x, True), g["snseq/y0"]) < TOL
    assert relerr(g["snseq/u2"], g["snseq/u3"]) == 0.0  # eval call leaves the buffers alone


def _models(tag, cfg, dtype):
    keys = load_keys(cfg)
    au = filled_sd(keys["au"], tag + "/au/", dtype)
    im = filled_sd(keys["im"], tag + "/im/", dtype)
    go.set_requires_grad(au)
    go.set_requires_grad(im)
    return au, im


def _check_grad_norms(sd, ref, gtol, floor_frac=1e-6):
    """Per-tensor gradient L2 norms; tensors whose gradient is mathematically zero (a conv bias in
    front of an instance/ada-in norm) hold rounding noise, hence the floor."""
    floor = floor_frac * max(ref.values())
    for k, v in ref.items():
        assert abs(float(sd[k].grad.norm()) - v) <= gtol * max(v, floor), k


def _check_nets(tag, cfg, dtype, tol, gtol, floor_frac=1e-6, use_img_att=False):
    g = load_npz("nets_%s.npz" % tag)
    meta = load_json("nets_%s.json" % tag)
    c = meta["config"]
    au, im = _models(tag, cfg, dtype)
    leaked, real, si, z = episode(tag, c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"], dtype)
    loss, fake, out = go.impersonator_forward(au, im, leaked, si, c["n"], z, True, True, use_img_att=use_img_att)
    assert relerr(loss, g["g/loss"]) < tol
    assert relerr(out, g["g/out"]) < tol
    gf = g["g/fake"]
    assert relerr(fake[:gf.shape[0], :gf.shape[1]], gf) < tol
    loss.mean().backward()
    _check_grad_norms(im, meta["meta"]["g/im_grad_norms"], gtol, floor_frac)
    _check_grad_norms(au, meta["meta"]["g/au_grad_norms"], gtol, floor_frac)
    for k in g.files:
        if k.startswith("g/grad/"):
            assert relerr(im[k[7:]].grad, g[k]) < gtol, k
    for sd in (au, im):
        for p in sd.values():
            p.grad = None
    res = go.authenticator_forward(au, fake.detach(), real, si, True, 0.0)
    for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake"]):
        assert relerr(res[i], g["d/" + nm]) < tol or float(np.abs(g["d/" + nm]).max()) == 0.0, nm
    assert (res[6].numpy() == g["d/pred_real"]).all() and (res[7].numpy() == g["d/pred_fake"]).all()
    res[0].mean().backward()
    _check_grad_norms(au, meta["meta"]["d/au_grad_norms"], gtol, floor_frac)
    for k in g.files:
        if k.startswith("d/grad/"):
            assert relerr(au[k[7:]].grad, g[k]) < gtol, k
        if k.startswith("d/buf/"):
            assert relerr(au[k[6:]], g[k]) < tol, k


def test_tiny_nets_fp64():
    _check_nets("tiny64", "16_1_32", torch.float64, 1e-9, 1e-7)


def test_tiny_nets_m2_fp64():
    _check_nets("tiny_m2", "16_1_32", torch.float64, 1e-9, 1e-7)


def test_tiny_nets_img_att_fp64():
    """use_img_att=True (ImgAttention branch, models/gim_img_models.py:391-396)."""
    _check_nets("tiny_att", "16_1_32", torch.float64, 1e-9, 1e-7, use_img_att=True)


def test_trainer_protocol_reg0():
    _check_trainer("reg0")


def test_trainer_protocol_reg10():
    _check_trainer("reg10")


def _check_trainer(tag):
    g = load_npz("trainer_%s.npz" % tag)
    meta = load_json("trainer_%s.json" % tag)
    c = meta["config"]
    au, im = _models(tag, "16_1_32", torch.float64)
    tr = go.OracleTrainer(au, im, c["n"], c["au_lr"], c["im_lr"], c["noise_lr"], reg_param=c["reg_param"])
    for it in range(3):
        leaked, real, si, z = episode("%s/it%d" % (tag, it), c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"])
        # MultiStepLR is built with last_epoch=-1 (its constructor steps once) and stepped BEFORE the
        # optimiser every iteration (gim_img_training.py:216): its epoch counter at iteration it is it+1
        scale = c["gamma"] if it + 1 >= c["milestones"][0] else 1.0
        assert abs(meta["meta"]["lrs"][it][0] - c["au_lr"] * scale) < 1e-15
        tr.im_opt.zero_grad()
        tr.im_training = True
        loss, fake, out = go.impersonator_forward(au, im, leaked, si, c["n"], z, tr.au_training, True)
        loss.mean().backward()
        tr.im_opt.step(lr_scale=scale)
        assert relerr(loss.mean(), g["it%d/g_loss" % it]) < 1e-8, it
        assert relerr(fake, g["it%d/fake" % it]) < 1e-8, it
        assert relerr(out, g["it%d/g_out" % it]) < 1e-8, it
        tr.au_training = True
        tr.au_opt.zero_grad()
        res = go.authenticator_forward(au, fake.detach(), real, si, True, c["reg_param"])
        res[0].mean().backward()
        tr.au_opt.step(lr_scale=scale)
        for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake"]):
            ref = g["it%d/d_%s" % (it, nm)]
            got = res[i].detach().mean()  # au_train_step returns means (gim_img_training.py:181-183)
            assert relerr(got, ref) < 1e-7 or float(np.abs(ref).max()) == 0.0, (it, nm)
    for k, (s, nrm) in meta["meta"]["au_final"].items():
        assert abs(float(au[k].detach().norm()) - nrm) <= 1e-7 * max(nrm, 1e-12), k
    for k, (s, nrm) in meta["meta"]["im_final"].items():
        assert abs(float(im[k].detach().norm()) - nrm) <= 1e-7 * max(nrm, 1e-12), k
    # eval pass: no power iteration, no grads
    leaked, real, si, z = episode(tag + "/eval", c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"])
    with torch.no_grad():
        # im_eval_step only flips the impersonator to eval(): the authenticator is still in train
        # mode there (gim_img_training.py:78), au_eval_step then flips it (:87)
        loss, fake, out = go.impersonator_forward(au, im, leaked, si, c["n"], z, True, False)
        res = go.authenticator_forward(au, fake, real, si, False, c["reg_param"], grad=False)
    assert relerr(loss.mean(), g["eval/g_loss"]) < 1e-7
    assert relerr(out, g["eval/g_out"]) < 1e-7
    assert relerr(res[0].mean(), g["eval/d_loss"]) < 1e-7


def test_bench_shape_32_fp64():
    _check_nets("om32_f64", "32_1_512", torch.float64, 1e-8, 1e-6)


def test_config5_shape_128_fp64():
    """128x128x3 (BASELINE config 5 shape), m=2 leaked images, fp64."""
    _check_nets("vox128_f64", "128_3_512", torch.float64, 1e-8, 1e-6)


def test_bench_shape_64_fp32_tolerance():
    """fp32 oracle vs the reference's own fp32 run at the 64x64x3 benchmark shape:
    within the 1e-3 tolerance north_star states for losses/logits."""
    _check_nets("vox64_f32", "64_3_512", torch.float32, 1e-3, 5e-2, floor_frac=1e-3)


@pytest.mark.parametrize("tag,reg", [("gauss", 0.0), ("gauss_r1", 1.0)])
def test_gaussian_toy_game_trainer(tag, reg):
    """BASELINE config 1 (d=10, m=1, n=5, k=10): 5 iterations of the reference's Gaussian trainer, with and without R1."""
    g = load_npz("gaussian.npz")
    meta = load_json("gaussian.json")
    c = meta["config"]
    au = filled_sd(meta["keys"]["au"], tag + "/au/")
    im = filled_sd(meta["keys"]["im"], tag + "/im/")
    tr = go.OracleGaussianTrainer(au, im, c["n"], c["au_lr"], c["im_lr"], reg_param=reg)
    for it in range(5):
        mu = pf.normal("%s/it%d/mu" % (tag, it), (c["B"], 1, c["d"]))
        smp = lambda nm, t: T(mu + c["sigma"] * pf.normal("%s/it%d/%s" % (tag, it, nm), (c["B"], t, c["d"])))  # noqa: E731
        leaked, real, si = smp("leaked", c["m"]), smp("real", c["n"]), smp("si", c["k"])
        z = T(pf.normal("%s/it%d/z" % (tag, it), (c["B"], c["n"], c["d"])))
        g_loss, fake, out, d_loss, out_real, out_fake = tr.step(leaked, real, si, z)
        assert relerr(g_loss.mean(), g["%s/it%d/g_loss" % (tag, it)]) < 1e-9
        assert relerr(fake, g["%s/it%d/fake" % (tag, it)]) < 1e-9
        assert relerr(out, g["%s/it%d/g_out" % (tag, it)]) < 1e-9
        assert relerr(d_loss.mean(), g["%s/it%d/d_loss" % (tag, it)]) < 1e-9
        assert relerr(out_real.mean(), g["%s/it%d/d_out_real" % (tag, it)]) < 1e-9
    for kk in au:
        assert relerr(au[kk], g["%s/final/au/%s" % (tag, kk)]) < 1e-9, kk
    for kk in im:
        if not kk.startswith("out_mlp"):
            assert relerr(im[kk], g["%s/final/im/%s" % (tag, kk)]) < 1e-9, kk


# ------------------------------------------------------------------------------------------------------------------
# round-2 fixtures: sub-networks on their own, gradient TENSOR samples at a benchmark shape, the dataset sample contract
# ------------------------------------------------------------------------------------------------------------------
SUBNETS = {
    "encoder": (dict(x=(3, 1, 16, 16)), lambda sd, x: go.encoder(sd, "", x, True)),
    "envdecoder": (dict(x=(3, 32)), lambda sd, x: go.env_decoder(sd, "", x, True)),
    "img2img": (dict(x=(3, 2, 16, 16), style=(3, 32)), lambda sd, x, style: go.img2img(sd, "", x, style, True)),
}


@pytest.mark.parametrize("name", sorted(SUBNETS))
def test_subnet_matches_reference(name):
    """Encoder / EnvDecoder / AdaInImage2Image alone: the oracle against the reference's fp64 run."""
    g = load_npz("subnets.npz")
    inputs, fn = SUBNETS[name]
    ks = [(k[len(name) + 3:], g[k].shape) for k in g.files if k.startswith(name + "/g/") or k.startswith(name + "/b/")]
    sd = filled_sd(ks, name + "/")
    go.set_requires_grad(sd)
    xs = {k: T(pf.normal("%s/%s" % (name, k), s)).requires_grad_() for k, s in inputs.items()}
    y = fn(sd, **xs)
    assert relerr(y, g[name + "/y"]) < 1e-9
    (y * T(pf.uniform(name + "/r", tuple(y.shape)))).sum().backward()
    for k, x in xs.items():
        assert relerr(x.grad, g["%s/d_%s" % (name, k)], atol=1e-9) < 1e-7, k
    gmax = max(float(np.linalg.norm(g[k])) for k in g.files if k.startswith(name + "/g/"))
    for k in g.files:
        if k.startswith(name + "/g/"):
            assert relerr(sd[k[len(name) + 3:]].grad, g[k], atol=1e-9 * gmax) < 1e-6, k
        if k.startswith(name + "/b/"):
            assert relerr(sd[k[len(name) + 3:]], g[k]) < TOL, k


def test_bench_shape_32_gradient_tensor_samples():
    """nets_om32_f64_grads.npz: strided samples of whole gradient tensors (G step and D step, 32x32x1, style 512) from the
    reference - position-sensitive, unlike the per-tensor norms of nets_om32_f64.json."""
    tag = "om32_f64"
    gs = load_npz("nets_%s_grads.npz" % tag)
    c = load_json("nets_%s.json" % tag)["config"]
    au, im = _models(tag, "32_1_512", torch.float64)
    leaked, real, si, z = episode(tag, c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"], torch.float64)
    loss, fake, out = go.impersonator_forward(au, im, leaked, si, c["n"], z, True, True)
    loss.mean().backward()
    n = 0

    def check_all(sd, prefix):   # (round 3) thin samples of EVERY parameter gradient; zero gradients compare against the pass's scale
        keys = [k for k in gs.files if k.startswith(prefix)]
        gmax = max(float(np.linalg.norm(gs[k].astype(np.float64))) for k in keys)
        for k in keys:
            _, stride, name = k.split("/", 2)
            assert relerr(sd[name].grad.reshape(-1)[::int(stride)], gs[k], atol=1e-7 * gmax) < 1e-6, k
        return len(keys)
    for k in gs.files:
        if k.startswith("g/"):
            _, stride, name = k.split("/", 2)
            assert relerr(im[name].grad.reshape(-1)[::int(stride)], gs[k]) < 1e-6, k   # the fixture stores float32
            n += 1
    n_all = check_all(im, "g_all/")
    for sd in (au, im):
        for p in sd.values():
            p.grad = None
    go.authenticator_forward(au, fake.detach(), real, si, True, 0.0)[0].mean().backward()
    for k in gs.files:
        if k.startswith("d/"):
            _, stride, name = k.split("/", 2)
            assert relerr(au[name].grad.reshape(-1)[::int(stride)], gs[k]) < 1e-6, k
            n += 1
    n_all += check_all(au, "d_all/")
    assert n == 10 and n_all == len(gs.files) - 10 and n_all > 300


def test_dataset_contract_fixture_pins_the_numpy_restatement():
    """tests/golden/data.npz (the reference's ImgGIMDataSet on PNG files of the bank): the value map the GPU gather kernel is
    tested against - float32(u8) / 255 * 2 + (-1), horizontal flip along W - reproduces the reference's tensors bit for bit,
    and the container contract holds (class filter, index -> class, distinct images of one class, set sizes)."""
    g = load_npz("data.npz")
    meta = load_json("data.json")
    c = meta["config"]
    bank, offs = g["bank"], g["offsets"]
    assert int(g["n_classes"]) == sum(1 for s in c["sizes"] if s >= c["m"] + c["n"] + c["k"]) == 3
    assert int(g["len"]) == 3 * c["example_cnt_per_class"]
    for e, ex in enumerate(meta["examples"]):
        assert ex["class"] == ex["index"] // c["example_cnt_per_class"]
        seen = []
        for part, t in (("leaked_sample", c["m"]), ("real_sample", c["n"]), ("si_sample", c["k"])):
            ref = g["ex%d/%s" % (e, part)]
            assert ref.shape == (t, c["C"], c["S"], c["S"]) and ref.dtype == np.float32 and np.abs(ref).max() <= 1.0
            for i, (src, flip) in enumerate(zip(g["ex%d/%s/src" % (e, part)], g["ex%d/%s/flip" % (e, part)])):
                img = bank[src][:, ::-1] if flip else bank[src]
                mine = (img.astype(np.float32) / np.float32(255.0)) * np.float32(2.0) + np.float32(-1.0)
                assert np.array_equal(mine.transpose(2, 0, 1), ref[i]), (e, part, i)
                seen.append(int(src))
        cls = ex["bank_class"]
        assert len(set(seen)) == len(seen) == c["m"] + c["n"] + c["k"] and all(offs[cls] <= v < offs[cls + 1] for v in seen)


def test_gaussian_caller_loop_and_game_value_vs_reference_golden():
    """BASELINE config 1's caller: the reference's own train() (training/gim_gaussian_training.py:50-151) run for 5 iterations
    from a seeded default generator, float32 (tests/golden/gaussian_loop.json: every logger call).  The oracle's restatement
    consumes the generator in the same order - mu, real, leaked, si, then z - and logs the same stream; and its closed-form game
    value equals the reference's theory/theoretic_game_value.py:10-20 on seven (m, n, d, k) points (SURVEY.md section 4 KATs)."""
    meta = load_json("gaussian_loop.json")
    g = load_npz("gaussian_loop.npz")
    c = meta["config"]
    keys = load_json("gaussian.json")["keys"]
    au = filled_sd(keys["au"], "gauss_loop/au/", torch.float32)
    im = filled_sd(keys["im"], "gauss_loop/im/", torch.float32)
    tr = go.OracleGaussianTrainer(au, im, c["n"], c["au_lr"], c["im_lr"], reg_param=c["reg_param"], m=c["m"], k=c["k"])
    torch.manual_seed(c["seed"])
    log = go.gaussian_train(tr, c["n_iters"], c["B"], c["d"], c["src_sigma"], c["prior_sigma"], c["save_stats_every"])
    assert [(a, b, s_) for a, b, s_, _ in log] == [(a, b, s_) for a, b, s_, _ in meta["scalars"]]
    for (cat, key, step, v), (_, _, _, ref) in zip(log, meta["scalars"]):
        assert abs(v - ref) <= 2e-5 * abs(ref) + 1e-6, (cat, key, step, v, ref)
    for kk in au:
        assert relerr(au[kk], g["final/au/" + kk]) < 1e-5, kk
    for kk in im:
        if kk.startswith("out_mlp"):
            continue
        if kk == "env_noise_mapper.model.0.bias":
            # remove_noise_mean subtracts the set mean of the mapper's output: its (only) bias cancels, the gradient is rounding
            # noise and Adam turns noise into steps of ~lr - a random walk in the reference's run and in any other
            assert float((im[kk].detach().double() - T(g["final/im/" + kk])).abs().max()) <= c["n_iters"] * c["im_lr"], kk
            continue
        assert relerr(im[kk], g["final/im/" + kk]) < 1e-5, kk
    assert meta["saves"] == [s_ for s_ in range(c["n_iters"]) if s_ % c["save_every"] == 0]
    kats = meta["game_value_mnk"]
    assert len(kats) == 7 and [1, 5, 10, 10, 0.9211306086938573] in kats
    for m_, n_, d_, k_, ref in kats:
        assert abs(go.game_value_mnk(m_, n_, d_, k_) - ref) < 1e-14, (m_, n_, d_, k_)
