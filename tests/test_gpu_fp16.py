"""GPU parity of the opt-in fp16 matrix path (BASELINE config 5 "fp16 MFMA"; csrc/conv_f16.inc, ops.set_matrix_path("fp16")):
operands rounded to fp16 when staged into LDS, v_mfma_f32_32x32x16_f16, fp32 accumulation.

Two kinds of bound.  (1) KERNEL correctness, tight: products of fp16 values are exact in fp32, so against an fp64 reference
computed from operands ROUNDED TO FP16 THE WAY THE KERNELS ROUND THEM (after the fused LeakyReLU; weights as stored) the fp16
kernels must agree like the fp32 ones do - 3e-5, the accumulation order is all that differs.  (2) What the ROUNDING itself costs,
loose: against the un-rounded fp64 reference a convolution's output moves by ~4e-4 of its norm (two operands, half an ulp of
2^-11 each, independent per product); the folded forms - whose folded weights are rounded AFTER folding - are checked that way."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gim_oracle as go
from oracle import portable_fill as pf
from tests.helpers import T, relerr

pytestmark = pytest.mark.gpu

TOL = 3e-5          # fp32-accumulate vs fp64 on identical (fp16-representable) operands
TOL_ROUND = 2e-3    # 5x the ~4e-4 a conv output moves when both operands are rounded to fp16


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def nhwc(x):
    return x.detach().permute(0, 2, 3, 1).contiguous().float().to(dev())


def nchw(y):
    return y.detach().permute(0, 3, 1, 2).double().cpu()


def cl_weight(w):
    return w.detach().float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_()


def r16(t):
    """Round to fp16 (nearest even) and back: what v_cvt_pk_f16_f32 does to an operand."""
    return t.detach().float().half().double()


@pytest.fixture()
def fp16_path():
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    prev = ops.set_matrix_path("fp16")
    yield ops
    ops.set_matrix_path(prev)


PLAIN = [
    # N, Cin, Cout, K, H, slope, res
    (3, 64, 64, 3, 16, 0.2, True),       # 64 x 64 tiles
    (2, 128, 256, 3, 32, 0.2, False),    # 64 x 128 tiles
    (5, 32, 96, 3, 8, 1.0, True),        # ragged output channels (96 of a 128-wide tile), one K step per tap
    (4, 64, 128, 1, 16, 1.0, False),     # 1 x 1
    (2, 32, 64, 9, 16, 0.2, False),      # 9 x 9: 81 taps
    (40, 64, 128, 3, 32, 0.2, True),     # M = 40960: 128 x 128 tiles, several pixel slices in the weight gradient
    (70, 128, 128, 3, 2, 0.2, False),    # 2 x 2 maps: split-K forward, generic (non-FASTB) gather in the weight gradient
]


@pytest.mark.parametrize("case", PLAIN, ids=[str(c) for c in PLAIN])
def test_fp16_plain_conv_vs_fp16_rounded_reference(case, fp16_path):
    ops = fp16_path
    N, Cin, Cout, K, H, slope, use_res = case
    tag = "f16conv%s" % (case,)
    x = T(pf.normal(tag + "x", (N, Cin, H, H)))
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K))
    b = T(pf.normal(tag + "b", (Cout,)))
    res = T(pf.normal(tag + "r", (N, Cout, H, H))) if use_res else None
    sig = 1.7
    # the plan says fp16 for all three launches (forward, dgrad on transposed weights, weight gradient)
    import ctypes
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    sh = ops._shape(N, H, H, Cin, Cout, K, 0, slope)
    for kind in (0, 2, 3):
        out = (ctypes.c_int32 * 8)()
        assert _lib.load().gim_conv_launch_plan(ctypes.byref(sh), kind, ctypes.cast(out, ctypes.c_void_p)) == 0
        assert out[7] == 2, ("launch kind %d is not on the fp16 path" % kind, list(out))
    # reference on fp16-rounded operands: activated input, weights, incoming gradient
    xa = r16(F.leaky_relu(x, slope)).requires_grad_()
    wr = r16(w).requires_grad_()
    y = F.conv2d(xa, wr, None, padding=(K - 1) // 2) / sig + b.view(1, -1, 1, 1)
    if use_res:
        y = y + res
    dy = T(pf.uniform(tag + "dy", tuple(y.shape)))
    dxa, dw = torch.autograd.grad(F.conv2d(xa, wr, None, padding=(K - 1) // 2) / sig, (xa, wr), r16(dy))
    dx = dxa * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, slope))
    db = dy.sum((0, 2, 3))

    xg = nhwc(x).requires_grad_()
    wg = cl_weight(w)
    bg = b.float().to(dev()).requires_grad_()
    rg = nhwc(res).requires_grad_() if use_res else None
    sg = torch.tensor([sig], device=dev())
    u0, v0 = torch.zeros(Cout, device=dev()), torch.zeros(Cin * K * K, device=dev())   # sigma's own gradient term off (test_sn_conv covers it)
    yg = ops.conv2d(xg, wg, bg, rg, sg, u0, v0, 0, slope)
    assert relerr(nchw(yg), y) < TOL, "forward"
    (yg * nhwc(dy)).sum().backward()
    assert relerr(nchw(xg.grad), dx) < TOL, "dx"
    assert relerr(wg.grad.double().cpu(), dw) < TOL, "dw"
    assert relerr(bg.grad.double().cpu(), db) < TOL, "db (fp32: summed before the operands are rounded)"
    # and what the rounding costs against the un-rounded reference
    y0 = F.conv2d(F.leaky_relu(x, slope), w, None, padding=(K - 1) // 2) / sig + b.view(1, -1, 1, 1) + (res if use_res else 0)
    e = relerr(nchw(yg), y0)
    print("fp16 operands vs un-rounded fp64, forward: %.2e" % e)
    assert 1e-6 < e < TOL_ROUND


FOLDED = [
    # N, Cin, Cout, K, H, pool, ups
    (3, 64, 64, 3, 16, True, False),      # avg-pool folded: stride-2, 4 x 4 taps
    (2, 64, 128, 3, 32, True, False),
    (2, 32, 64, 9, 16, True, False),      # 10 x 10 folded taps
    (3, 64, 32, 3, 16, False, True),      # sub-pixel form of conv(up2(x)): 4 parity classes of 2 x 2 taps
    (2, 256, 128, 3, 32, False, True),
]


@pytest.mark.parametrize("case", FOLDED, ids=[str(c) for c in FOLDED])
def test_fp16_folded_conv_vs_fp64(case, fp16_path):
    ops = fp16_path
    N, Cin, Cout, K, H, pool, ups = case
    tag = "f16fold%s" % (case,)
    Hs = H >> (1 if ups else 0)
    x = T(pf.normal(tag + "x", (N, Cin, Hs, Hs))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    xa = F.leaky_relu(x, 0.2)
    if ups:
        xa = go.upsample2(xa)
    y = F.conv2d(xa, w, b, padding=(K - 1) // 2)
    if pool:
        y = F.avg_pool2d(y, 2)
    dy = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * dy).sum().backward()
    xg = nhwc(x).requires_grad_()
    wg = cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    yg = ops.conv2d(xg, wg, bg, None, None, None, None, 1 if ups else 0, 0.2, pool)
    (yg * nhwc(dy)).sum().backward()
    errs = (relerr(nchw(yg), y), relerr(nchw(xg.grad), x.grad), relerr(wg.grad.double().cpu(), w.grad), relerr(bg.grad.double().cpu(), b.grad))
    print("fp16 folded conv vs fp64: y %.2e dx %.2e dw %.2e db %.2e" % errs)
    assert all(e < TOL_ROUND for e in errs[:3]) and errs[3] < TOL, errs
    assert errs[0] > 1e-6, "the fp16 path did not run"


def test_fp16_path_leaves_ineligible_launches_on_fp32(fp16_path):
    """Image layers (3 / 6 input channels), < 32 output channels and linears stay on the fp32 MFMA under the switch: bit-identical
    to the default path (compared in the deterministic mode: a default launch that splits K adds its slices in a varying order)."""
    ops = fp16_path
    prev_det = ops.set_deterministic(True)
    for (N, Cin, Cout, K, H) in ((2, 3, 64, 3, 16), (2, 64, 16, 1, 16), (2, 64, 3, 9, 16)):
        tag = "f16inel%s" % ((N, Cin, Cout, K, H),)
        xg = nhwc(T(pf.normal(tag + "x", (N, Cin, H, H))))
        wg = cl_weight(T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)))
        y16 = ops.conv2d(xg, wg, None, None, None, None, None, 0, 0.2)
        ops.set_matrix_path("fp32")
        y32 = ops.conv2d(xg, wg, None, None, None, None, None, 0, 0.2)
        ops.set_matrix_path("fp16")
        assert torch.equal(y16, y32), (Cin, Cout, K)
    xl = torch.randn(8, 64, device=dev())
    wl = torch.randn(32, 64, device=dev())
    y16 = ops.linear(xl, wl)
    ops.set_matrix_path("fp32")
    assert torch.equal(y16, ops.linear(xl, wl))
    ops.set_matrix_path("fp16")
    ops.set_deterministic(prev_det)


def _measure_nets(tag, cfg):
    """One generator step and one discriminator step on the reference fixture nets_<tag> (fp64 golden): relative errors of the
    losses, logits, fake images and the per-tensor gradient norms (worst and median over the parameters whose gradient is not
    mathematically zero).  The backward passes run the way the step functions run them (loss scale of the matrix path in force)."""
    import tempfile
    import optimalstrategiesagainstgenerativeattacks_amd as G
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    from tests.helpers import episode, load_json, load_npz
    from tests.test_gpu_models import _product_models
    g, meta = load_npz("nets_%s.npz" % tag), load_json("nets_%s.json" % tag)
    c = meta["config"]
    au, im = _product_models(tag, cfg)
    leaked, real, si, z = [t.float().to(dev()) for t in episode(tag, c["B"], c["m"], c["n"], c["k"], c["c"], c["s"], c["d"])]
    with tempfile.TemporaryDirectory() as td:
        tr = G.GIMImgTrainer(td, c["m"], c["n"], c["k"], au, im, 1e-4, 1e-4, 1e-6, reg_param=0.0)
    au.train(); im.train()
    S = ops.loss_scale()
    out = {}
    tr.impersonator_opt.zero_grad()
    loss, fake, logits = tr.forward(mode="impersonator_forward", leaked_sample=leaked, si_sample=si, z=z)
    gf = g["g/fake"]
    out["g_loss"], out["g_logits"], out["fake"] = relerr(loss, g["g/loss"]), relerr(logits, g["g/out"]), relerr(fake[:gf.shape[0], :gf.shape[1]], gf)
    (loss.mean() * S).backward()

    def norm_errs(mod, ref):
        gmax = max(ref.values())
        es = [abs(float(p_.grad.double().norm()) / S - ref[k_]) / ref[k_] for k_, p_ in mod.named_parameters()
              if k_ in ref and p_.grad is not None and ref[k_] > 1e-3 * gmax]
        return float(np.max(es)), float(np.median(es))
    out["g_gradnorm_worst"], out["g_gradnorm_median"] = norm_errs(im, meta["meta"]["g/im_grad_norms"])
    tr.authenticator_opt.zero_grad()
    res = tr.forward(mode="authenticator_forward", fake_sample=fake.detach(), real_sample=real, si_sample=si)
    for i, nm in enumerate(["loss", "loss_real", "loss_fake", "reg", "out_real", "out_fake"]):
        if float(np.abs(g["d/" + nm]).max()) > 0:
            out["d_" + nm] = relerr(res[i], g["d/" + nm])
    out["preds_equal"] = bool((res[6].cpu().numpy() == g["d/pred_real"]).all() and (res[7].cpu().numpy() == g["d/pred_fake"]).all())
    (res[0].mean() * S).backward()
    out["d_gradnorm_worst"], out["d_gradnorm_median"] = norm_errs(au, meta["meta"]["d/au_grad_norms"])
    return out


@pytest.mark.parametrize("tag,cfg", [("vox128_m5n20k20", "128_3_512"), ("vox64_f64", "64_3_512"), ("tiny64", "16_1_32")])
def test_fp16_accuracy_gate_on_reference_fixtures(tag, cfg, fp16_path):
    """The ACCURACY GATE of the fp16 experiment, measured on the reference's fp64 fixtures - first of all
    BASELINE config 5 as stated (nets_vox128_m5n20k20: 128 x 128 x 3, m = 5, n = 20, k = 20): north_star's 1e-3 on losses and
    logits.  The numbers are printed whatever they are (DESIGN.md section 8 records them); the asserts are 3x what was measured
    on an MI355X, i.e. they pin the path's behaviour, they do not claim the gate."""
    ops = fp16_path
    e16 = _measure_nets(tag, cfg)
    ops.set_matrix_path("fp32")
    e32 = _measure_nets(tag, cfg)
    ops.set_matrix_path("fp16")
    print("fixture %s, relative errors vs the fp64 reference [fp16 path | fp32 path]:" % tag)
    for k_ in e16:
        if k_ != "preds_equal":
            print("   %-20s %.2e | %.2e" % (k_, e16[k_], e32[k_]))
    print("   gate (losses and logits <= 1e-3): %s" % ("PASSED" if max(e16[k_] for k_ in e16 if k_.endswith(("loss", "logits", "out_real", "out_fake", "loss_real", "loss_fake"))) <= 1e-3 else "FAILED"))
    assert e16["preds_equal"]
    for k_ in ("g_loss", "g_logits", "fake", "d_loss", "d_out_real", "d_out_fake"):
        assert e16[k_] < 2e-2, (k_, e16[k_])
    assert e16["g_gradnorm_worst"] < 0.5 and e16["d_gradnorm_worst"] < 0.5, e16
