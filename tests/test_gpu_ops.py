"""GPU parity, operator level: each C-ABI entry point (through the autograd operators of ops.py) against
fp64 CPU restatements (torch eager / oracle/gim_oracle.py) on seeded inputs.  Tolerances are fp32-vs-fp64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gim_oracle as go
from oracle import portable_fill as pf
from tests.helpers import T, relerr, relerr_floor

pytestmark = pytest.mark.gpu

TOL = 3e-5


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def nhwc(x):  # NCHW cpu f64 -> NHWC cuda f32
    return x.detach().permute(0, 2, 3, 1).contiguous().float().to(dev())


def nchw(y):  # NHWC cuda -> NCHW cpu f64
    return y.detach().permute(0, 3, 1, 2).double().cpu()


def cl_weight(w):  # [Cout,Cin,k,k] cpu f64 -> cuda f32 parameter stored channels-last
    return w.detach().float().to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_()


CONV_CASES = [
    # N, Cin, Cout, K, H, ups, slope, res, sigma
    (2, 16, 32, 3, 8, 0, 1.0, False, False),
    (2, 16, 64, 3, 8, 0, 0.2, True, True),
    (3, 32, 128, 3, 16, 0, 0.2, False, True),      # 128x128 tile path (M=768 -> 64x64 tiles)
    (9, 64, 160, 3, 64, 0, 1.0, False, True),      # big M: 128x128 tiles, Cout not multiple of tile
    (2, 3, 64, 3, 16, 0, 0.2, False, True),        # generic-K (Cin=3)
    (2, 6, 64, 9, 16, 0, 0.2, False, True),        # 9x9 generic-K
    (2, 64, 3, 9, 16, 1, 0.2, True, True),         # small Cout, upsample, 9x9
    (2, 1, 32, 1, 8, 0, 1.0, False, True),         # 1x1, Cin=1
    (4, 32, 16, 1, 4, 1, 1.0, False, True),        # 1x1 upsample
    (5, 48, 48, 3, 2, 1, 0.2, True, True),         # tiny maps (1x1 -> 2x2)
    (70, 128, 128, 3, 1, 0, 1.0, False, False),    # 1x1 spatial with 3x3 kernel (only centre tap valid)
    (2, 128, 256, 3, 32, 0, 0.2, False, True),     # M=2048: 128-wide tiles
    (3, 32, 64, 3, 16, 1, 0.2, False, True),       # sub-pixel form, vector paths
    (2, 16, 16, 9, 8, 1, 1.0, True, False),        # sub-pixel 9x9 (5x5 taps per class)
    (1, 16, 32, 3, 128, 0, 0.2, True, True),       # patch-resident loop, 128-wide map: a 64-pixel tile is half an image row
    (5, 32, 48, 3, 8, 0, 0.2, False, True),        # patch-resident loop, 8x8 map: one tile = one image, ragged output channels
    (3, 48, 96, 3, 4, 0, 0.2, True, True),         # 4x4 map: below the patch kernel's 64 pixels, tap-major loop
    # >= 32 images on maps of <= 256 pixels: position-major rows, padding taps skipped (csrc/conv_igemm.hip Geo.pm)
    (40, 32, 48, 3, 4, 0, 0.2, True, True),        # 4x4 map, 64-row tiles straddle pixel positions (40 images), ragged output channels
    (35, 16, 32, 3, 8, 0, 1.0, False, True),       # 35 images: positions do not start on 4-row groups
    (64, 32, 64, 3, 2, 0, 0.2, True, False),       # 2x2 map: one tile = one position, 4 of 9 taps valid
    (33, 32, 64, 3, 8, 1, 0.2, False, True),       # sub-pixel classes on a 4x4 -> 8x8 map
    (48, 32, 32, 3, 16, 1, 0.2, True, True),       # sub-pixel, residual at full resolution, 8x8 -> 16x16
    (32, 64, 128, 3, 16, 0, 0.2, True, True),      # 16x16 map (256 pixels, the largest position-major map); forward: patch-resident loop
]

POOL_CASES = [
    # N, Cin, Cout, K, H, slope, res
    (2, 16, 32, 3, 8, 0.2, True),
    (3, 64, 64, 3, 16, 0.2, True),
    (2, 3, 8, 3, 16, 0.2, False),       # generic-K
    (2, 16, 48, 9, 16, 0.2, True),      # 9x9 + pool = 10x10 stride 2
    (2, 6, 64, 9, 16, 1.0, False),
    (5, 128, 128, 3, 2, 0.2, True),     # 2x2 -> 1x1
    (9, 64, 160, 3, 32, 0.2, True),
    (3, 32, 64, 1, 16, 1.0, True),      # 1x1 + pool = 2x2 stride 2 (the skip convs of ResBlockDown)
    (2, 3, 64, 1, 16, 1.0, False),      # ... of the first block (image channels: generic-K)
    (5, 128, 256, 1, 2, 1.0, False),    # 2x2 -> 1x1
    # position-major rows (>= 32 images, small maps): the stride-2 fold and its parity-class dgrad skip their padding taps
    (40, 32, 32, 3, 8, 0.2, True),      # 8x8 -> 4x4, tiles straddle positions
    (36, 16, 48, 3, 4, 0.2, False),     # 4x4 -> 2x2
    (32, 32, 64, 3, 16, 1.0, True),     # 16x16 -> 8x8
    (34, 16, 16, 9, 8, 0.2, False),     # 9x9 + pool: 100 taps - more than the tap mask holds, stays image-major
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_fwd_bwd(case):
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    N, Cin, Cout, K, H, ups, slope, use_res, use_sigma = case
    tag = "conv%s" % (case,)
    Hs = H >> ups
    x = T(pf.normal(tag + "x", (N, Cin, Hs, Hs))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    res = T(pf.normal(tag + "r", (N, Cout, H, H))).requires_grad_() if use_res else None
    sig = 1.7 if use_sigma else 1.0
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    if ups:
        xa = go.upsample2(xa)
    y = F.conv2d(xa, w / sig, b, padding=(K - 1) // 2)
    if use_res:
        y = y + res
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()

    xg = nhwc(x).requires_grad_()
    wg = cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    rg = nhwc(res).requires_grad_() if use_res else None
    sg = torch.tensor([sig], device=dev()) if use_sigma else None
    yg = ops.conv2d(xg, wg, bg, rg, sg, None, None, ups, slope)
    assert relerr(nchw(yg), y) < TOL
    # sigma given without u/v: the spectral term of the weight gradient is skipped only when sigma is None, so
    # feed u = 0 to test the plain 1/sigma scaling of wgrad here (the spectral term is covered by test_sn_conv)
    if use_sigma:
        u0 = torch.zeros(Cout, device=dev())
        v0 = torch.zeros(Cin * K * K, device=dev())
        yg = ops.conv2d(xg, wg, bg, rg, sg, u0, v0, ups, slope)
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL, "dx"
    assert relerr(wg.grad.double().cpu(), w.grad) < TOL, "dw"
    assert relerr(bg.grad.double().cpu(), b.grad) < TOL, "db"
    if use_res:
        assert relerr(nchw(rg.grad), res.grad) < TOL, "dres"


TINY_CASES = [
    # N, C, K, H, W, slope, res (0 none, 1 full resolution, 2 half resolution), x stored activated
    (3, 3, 9, 64, 64, 0.2, 2, True),     # the generator's last 9x9 conv as the model calls it (models/gim_img_models.py:187-193)
    (2, 3, 3, 32, 32, 0.2, 2, True),     # the decoder's last 3x3 conv (models/gim_img_models.py:128)
    (2, 3, 9, 8, 8, 0.2, 1, False),      # a map smaller than the 16 x 16 tile
    (5, 3, 3, 16, 32, 1.0, 0, False),    # non-square, no activation
    (2, 1, 9, 32, 32, 0.2, 2, False),    # one-channel data (32x32x1 workload)
    (3, 1, 3, 4, 4, 0.2, 0, False),
]


@pytest.mark.parametrize("case", TINY_CASES, ids=[str(c) for c in TINY_CASES])
def test_tiny_image_conv_direct_kernels(case):
    """3 -> 3 / 1 -> 1 image layers run the direct kernels of conv_tiny.hip (launch plan: loop form 3) - forward with bias, 1/sigma and
    a full- or half-resolution residual, the masked input gradient, weight and bias gradients - against fp64 F.conv2d autograd."""
    import ctypes
    from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops
    N, C, K, H, W, slope, res_kind, x_act = case
    tag = "tiny%s" % (case,)
    x = T(pf.normal(tag + "x", (N, C, H, W))).requires_grad_()
    w = T(pf.normal(tag + "w", (C, C, K, K)) / np.sqrt(C * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (C,))).requires_grad_()
    res = None
    if res_kind == 1:
        res = T(pf.normal(tag + "r", (N, C, H, W))).requires_grad_()
    elif res_kind == 2:
        res = T(pf.normal(tag + "r", (N, C, H // 2, W // 2))).requires_grad_()
    sig = 1.3
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    y = F.conv2d(xa, w / sig, b, padding=(K - 1) // 2)
    if res is not None:
        y = y + (go.upsample2(res) if res_kind == 2 else res)
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()
    sh = ops._shape(N, H, W, C, C, K, 0, slope)
    for kind in (0, 1, 3):
        out = (ctypes.c_int32 * 8)()
        _lib.check(_lib.load().gim_conv_launch_plan(ctypes.byref(sh), kind, ctypes.cast(out, ctypes.c_void_p)), "plan")
        assert out[7] == 3, ("not the direct kernel", kind, list(out))
    # the model stores this conv's input ACTIVATED (its producer applied the LeakyReLU): feed lrelu(x) with x_act=True then
    xin = nhwc(F.leaky_relu(x, slope) if x_act else x).requires_grad_()
    wg = cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    rg = nhwc(res).requires_grad_() if res is not None else None
    sg = torch.tensor([sig], device=dev())
    u0, v0 = torch.zeros(C, device=dev()), torch.zeros(C * K * K, device=dev())
    yg = ops.conv2d(xin, wg, bg, rg, sg, u0, v0, 0, slope, False, res_kind == 2, None, None, x_act)
    assert relerr(nchw(yg), y) < TOL, "forward"
    (yg * nhwc(r)).sum().backward()
    # gradient w.r.t. the RAW x either way (an activated input hands back the raw gradient: ops.ConvFn)
    assert relerr(nchw(xin.grad), x.grad) < TOL, "dx"
    assert relerr(wg.grad.double().cpu(), w.grad) < TOL, "dw"
    assert relerr(bg.grad.double().cpu(), b.grad) < TOL, "db"
    if res is not None:
        assert relerr(nchw(rg.grad), res.grad) < TOL, "dres"


@pytest.mark.parametrize("N,Cin,Cout,K,H,slope,post", [(3, 64, 3, 9, 32, 0.2, 0.2), (2, 16, 4, 5, 8, 1.0, 1.0), (2, 32, 1, 9, 16, 0.2, 1.0),
                                                     (5, 24, 3, 13, 8, 0.2, 0.2), (1, 64, 3, 9, 2, 0.2, 1.0)])
def test_subpixel_conv_to_image_channels_stacked_classes(N, Cin, Cout, K, H, slope, post):
    """conv_KxK(up2(x)) to <= 4 channels, K = 5, 9, 13 (the generator's last 9x9 64->3 layer, models/gim_img_models.py:187-193): the
    forward runs the four output-parity classes stacked into one plain convolution + a depth-to-space copy
    (gim_conv2d_pack_subpixel_weights, gim_depth_to_space2), optionally storing the activated output; gradients: the sub-pixel forms."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    tag = "subpix%s" % ((N, Cin, Cout, K, H),)
    Hs = H // 2
    x = T(pf.normal(tag + "x", (N, Cin, Hs, Hs))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    sig = 1.3
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    y = F.conv2d(go.upsample2(xa), w / sig, b, padding=(K - 1) // 2)
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()

    xg, wg = nhwc(x).requires_grad_(), cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    sg = torch.tensor([sig], device=dev())
    u0, v0 = torch.zeros(Cout, device=dev()), torch.zeros(Cin * K * K, device=dev())
    sh = ops._shape(N, H, H, Cin, Cout, K, 1, slope, 0, 1, 0)
    assert ops._merged_subpixel(xg, wg, 1, None, sh)
    yg, act = ops.conv2d_post_act(xg, wg, bg, None, sg, u0, v0, 1, slope, post_slope=post)
    assert act == (post != 1.0)
    y_stored = F.leaky_relu(y, post) if act else y
    assert relerr(nchw(yg), y_stored.detach()) < TOL
    (yg * nhwc(r)).sum().backward()      # ConvFn's backward is that of the RAW output (its consumer un-does the activation)
    assert relerr(nchw(xg.grad), x.grad) < TOL, "dx"
    assert relerr(wg.grad.double().cpu(), w.grad) < TOL, "dw"
    assert relerr(bg.grad.double().cpu(), b.grad) < TOL, "db"
    # the four-class launch it replaces computes the same numbers
    old = ops._MERGED_SUBPIXEL
    ops._MERGED_SUBPIXEL = False
    try:
        y2 = ops.conv2d(xg.detach(), wg.detach(), bg.detach(), None, sg, None, None, 1, slope)
    finally:
        ops._MERGED_SUBPIXEL = old
    assert relerr(nchw(y2), y.detach()) < TOL


@pytest.mark.parametrize("case", POOL_CASES, ids=[str(c) for c in POOL_CASES])
def test_conv2d_pool_fold(case):
    """avgpool2(conv(lrelu(x))) + res as ONE stride-2 convolution with folded weights: fwd, dx, dw, db, dres."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    N, Cin, Cout, K, H, slope, use_res = case
    tag = "pool%s" % (case,)
    x = T(pf.normal(tag + "x", (N, Cin, H, H))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    res = T(pf.normal(tag + "r", (N, Cout, H // 2, H // 2))).requires_grad_() if use_res else None
    sig = 1.3
    xa = F.leaky_relu(x, slope) if slope != 1.0 else x
    y = F.avg_pool2d(F.conv2d(xa, w / sig, None, padding=(K - 1) // 2), 2) + b.view(1, -1, 1, 1)
    if use_res:
        y = y + res
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    wg = cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    rg = nhwc(res).requires_grad_() if use_res else None
    sg = torch.tensor([sig], device=dev())
    u0 = torch.zeros(Cout, device=dev())
    v0 = torch.zeros(Cin * K * K, device=dev())
    yg = ops.conv2d(xg, wg, bg, rg, sg, u0, v0, 0, slope, pool=True)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL, "dx"
    assert relerr(wg.grad.double().cpu(), w.grad) < TOL, "dw"
    assert relerr(bg.grad.double().cpu(), b.grad) < TOL, "db"
    if use_res:
        assert relerr(nchw(rg.grad), res.grad) < TOL, "dres"


@pytest.mark.parametrize("N,C1,C2,H,expect_act", [(6, 16, 64, 32, True), (2, 64, 64, 4, False)])
def test_conv_pair_with_activated_storage(N, C1, C2, H, expect_act):
    """conv_r1 -> LeakyReLU -> pooled conv_r2 (models/model_blocks.py:505-510) with the intermediate stored ACTIVATED by conv_r1's
    epilogue (ops.conv2d_post_act) and conv_r2 run with x_act: outputs and ALL gradients against fp64 autograd of the unfused
    form, in both regimes - a launch that can activate in its epilogue and a split-K launch that hands back the raw tensor."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    tag = "actpair%d_%d_%d_%d" % (N, C1, C2, H)
    x = T(pf.normal(tag + "x", (N, C1, H, H))).requires_grad_()
    w1 = T(pf.normal(tag + "w1", (C2, C1, 3, 3)) / np.sqrt(C1 * 9)).requires_grad_()
    b1 = T(pf.normal(tag + "b1", (C2,))).requires_grad_()
    w2 = T(pf.normal(tag + "w2", (C2, C2, 3, 3)) / np.sqrt(C2 * 9)).requires_grad_()
    b2 = T(pf.normal(tag + "b2", (C2,))).requires_grad_()
    h = F.conv2d(F.leaky_relu(x, 0.2), w1, b1, padding=1)
    y = F.avg_pool2d(F.conv2d(F.leaky_relu(h, 0.2), w2, b2, padding=1), 2)
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    w1g, w2g = cl_weight(w1), cl_weight(w2)
    b1g, b2g = (b.detach().float().to(dev()).requires_grad_() for b in (b1, b2))
    hg, act = ops.conv2d_post_act(xg, w1g, b1g, pre_slope=0.2, post_slope=0.2)
    assert act == expect_act, "which regime this shape was meant to exercise"
    if act:   # the stored intermediate is lrelu(h)
        assert relerr(nchw(hg), F.leaky_relu(h, 0.2)) < TOL
    else:
        assert relerr(nchw(hg), h) < TOL
    yg = ops.conv2d(hg, w2g, b2g, pre_slope=0.2, pool=True, x_act=act)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL, "dx"
    for nm, got, ref in (("dw1", w1g.grad, w1.grad), ("db1", b1g.grad, b1.grad), ("dw2", w2g.grad, w2.grad), ("db2", b2g.grad, b2.grad)):
        assert relerr(got.double().cpu(), ref) < TOL, nm


@pytest.mark.parametrize("N,Cin,Cout,H", [(2, 16, 24, 8), (40, 32, 48, 4), (33, 16, 64, 2)])   # the last two: position-major rows (Geo.pm)
def test_conv2d_residual_upsampled(N, Cin, Cout, H):
    """y = conv(x) + up2(res_low): residual stored at half resolution (skip of the up blocks)."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    K = 3
    x = T(pf.normal("ru/x%d" % N, (N, Cin, H, H))).requires_grad_()
    w = T(pf.normal("ru/w%d" % N, (Cout, Cin, K, K)) / 12).requires_grad_()
    res = T(pf.normal("ru/r%d" % N, (N, Cout, H // 2, H // 2))).requires_grad_()
    y = F.conv2d(x, w, None, padding=1) + go.upsample2(res)
    r = T(pf.uniform("ru/dy%d" % N, tuple(y.shape)))
    (y * r).sum().backward()
    xg, wg, rg = nhwc(x).requires_grad_(), cl_weight(w), nhwc(res).requires_grad_()
    yg = ops.conv2d(xg, wg, None, rg, None, None, None, 0, 1.0, res_ups=True)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL and relerr(wg.grad.double().cpu(), w.grad) < TOL
    assert relerr(nchw(rg.grad), res.grad) < TOL


@pytest.mark.parametrize("N,Cin,Cout,K,H,slope,x_act", [(3, 6, 64, 9, 32, 0.2, False), (4, 3, 64, 3, 16, 0.2, True), (2, 1, 32, 3, 16, 0.2, False),
                                                       (2, 6, 48, 9, 8, 1.0, False)])
def test_image_layer_rows_form_through_the_gradient_bucket(N, Cin, Cout, K, H, slope, x_act):
    """The row-contiguous form of the image layers (gim_pad_image + gim_conv2d_pack_rows_weights + gim_conv2d_fwd_rows, and
    gim_conv2d_wgrad_rows_acc with the batched finish's un-padding, fold code 3) as the TRAINING STEP runs it: weight and bias own
    pre-existing .grad buffers (the optimizer's flat bucket), so the weight gradient goes through the queue and is ADDED there, with
    the spectral-norm chain rule - against fp64 autograd of F.conv2d(lrelu(x), w / sigma(w)) + bias."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    assert ops._ROWS_FORM
    tag = "rows%s" % ((N, Cin, Cout, K, H, slope, x_act),)
    x = T(pf.normal(tag + "x", (N, Cin, H, H))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    u = T(pf.normal(tag + "u", (Cout,)))
    v = T(pf.normal(tag + "v", (Cin * K * K,)))
    u, v = u / u.norm(), v / v.norm()
    sigma = torch.dot(u, torch.mv(w.reshape(Cout, -1), v))          # differentiated w.r.t. w with u, v constant (spectral norm)
    y = F.conv2d(F.leaky_relu(x, slope) if slope != 1.0 else x, w / sigma, b, padding=(K - 1) // 2)
    r = T(pf.uniform(tag + "dy", tuple(y.shape)))
    (y * r).sum().backward()
    xin = nhwc(F.leaky_relu(x, slope) if x_act else x).requires_grad_()
    wg = cl_weight(w)
    bg = b.detach().float().to(dev()).requires_grad_()
    pre = 0.25
    wg.grad = torch.full_like(wg, pre)       # channels-last like the parameter: the bucket's memory order
    bg.grad = torch.full_like(bg, pre)
    sg = sigma.detach().float().reshape(1).to(dev())
    yg = ops.conv2d(xin, wg, bg, None, sg, u.float().to(dev()), v.float().to(dev()), 0, slope, x_act=x_act)
    assert relerr(nchw(yg), y) < TOL, "forward"
    (yg * nhwc(r)).sum().backward()
    assert not ops.wgrad_queue.jobs, "the queue was flushed at the end of backward"
    assert relerr(nchw(xin.grad), x.grad) < TOL, "dx"
    assert relerr((wg.grad - pre).double().cpu(), w.grad) < TOL, "dw (added into the existing buffer, through sigma)"
    assert relerr((bg.grad - pre).double().cpu(), b.grad) < TOL, "db"


@pytest.mark.parametrize("N,Cin,Cout,K,H,x_act", [(3, 32, 64, 3, 16, False), (2, 64, 128, 3, 32, True), (2, 16, 48, 3, 4, False), (5, 128, 128, 3, 2, True),
                                                 (2, 3, 32, 3, 16, False), (2, 6, 64, 9, 16, False), (3, 48, 32, 1, 8, False),
                                                 (40, 32, 64, 3, 4, True), (36, 16, 32, 3, 2, False)])   # position-major rows + the mask / pooled-skip epilogue
def test_conv_with_pooled_skip_reader_one_autograd_node(N, Cin, Cout, K, H, x_act):
    """ops.conv2d_forkpool: y = conv(lrelu(x)) and pooled = avgpool2(x) as ONE node whose backward adds the pooled branch's gradient in
    the dgrad epilogue (gim_conv2d_dgrad_res; patch-resident and tap-major kernels, split-K maps) - or, where the dgrad takes another
    launch form (image layers: x-fold), behind it - against fp64 autograd, and against the unfused pair of round 3."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    tag = "cfp%s" % ((N, Cin, Cout, K, H, x_act),)
    slope = 0.2
    x = T(pf.normal(tag + "x", (N, Cin, H, H))).requires_grad_()
    w = T(pf.normal(tag + "w", (Cout, Cin, K, K)) / np.sqrt(Cin * K * K)).requires_grad_()
    b = T(pf.normal(tag + "b", (Cout,))).requires_grad_()
    y = F.conv2d(F.leaky_relu(x, slope), w / 1.4, b, padding=(K - 1) // 2)
    pooled = F.avg_pool2d(x, 2)
    r1, r2 = T(pf.uniform(tag + "r1", tuple(y.shape))), T(pf.uniform(tag + "r2", tuple(pooled.shape)))
    ((y * r1).sum() + (pooled * r2).sum()).backward()
    sg = torch.tensor([1.4], device=dev())
    u0, v0 = torch.zeros(Cout, device=dev()), torch.zeros(Cin * K * K, device=dev())
    outs = []
    for fused in (True, False):
        prev = ops._FUSED_FORKPOOL
        ops._FUSED_FORKPOOL = fused
        try:
            xin = nhwc(F.leaky_relu(x, slope) if x_act else x).requires_grad_()
            wg = cl_weight(w)
            bg = b.detach().float().to(dev()).requires_grad_()
            yg, act, pg = ops.conv2d_forkpool(xin, wg, bg, sg, u0, v0, slope, None, 1.0, x_act, slope if x_act else 1.0)
            assert relerr(nchw(yg), y) < TOL and relerr(nchw(pg), pooled) < TOL and not act
            ((yg * nhwc(r1)).sum() + (pg * nhwc(r2)).sum()).backward()
        finally:
            ops._FUSED_FORKPOOL = prev
        assert relerr(nchw(xin.grad), x.grad) < TOL, ("dx", fused)
        assert relerr(wg.grad.double().cpu(), w.grad) < TOL and relerr(bg.grad.double().cpu(), b.grad) < TOL, fused
        outs.append(xin.grad)
    assert relerr(outs[0], outs[1]) < 1e-6


def test_linear_fwd_bwd():
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    for rows, din, dout, slope in [(5, 6, 10, 1.0), (80, 512, 1024, 0.2), (16, 320, 1, 0.2), (240, 96, 64, 1.0)]:
        tag = "lin%d_%d_%d" % (rows, din, dout)
        x = T(pf.normal(tag + "x", (rows, din))).requires_grad_()
        w = T(pf.normal(tag + "w", (dout, din)) / np.sqrt(din)).requires_grad_()
        b = T(pf.normal(tag + "b", (dout,))).requires_grad_()
        y = F.linear(F.leaky_relu(x, slope) if slope != 1.0 else x, w, b)
        r = T(pf.uniform(tag + "r", tuple(y.shape)))
        (y * r).sum().backward()
        xg = x.detach().float().to(dev()).requires_grad_()
        wg = w.detach().float().to(dev()).requires_grad_()
        bg = b.detach().float().to(dev()).requires_grad_()
        yg = ops.linear(xg, wg, bg, slope)
        assert relerr(yg, y) < TOL
        (yg * r.float().to(dev())).sum().backward()
        assert relerr(xg.grad, x.grad) < TOL
        assert relerr(wg.grad, w.grad) < TOL
        assert relerr(bg.grad, b.grad) < TOL


def test_grouped_linear_fwd_bwd():
    """ops.grouped_linear: n nn.Linear layers on one shared input in one launch (the generator's style projections,
    models/model_blocks.py:786-789,829-832) against fp64 F.linear autograd: outputs, the input gradient (sum over the layers),
    weight and bias gradients - returned, and ADDED into existing .grad buffers (the optimizer's bucket) where they exist.
    Widths cover the benchmark's (512 ... 64, 3) and ragged ones; one output is left without a gradient."""
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    rows, din = 80, 96
    widths = [64, 3, 130, 256, 16, 1]
    x = T(pf.normal("glin/x", (rows, din))).requires_grad_()
    ws = [T(pf.normal("glin/w%d" % g, (n_, din)) / np.sqrt(din)).requires_grad_() for g, n_ in enumerate(widths)]
    bs = [T(pf.normal("glin/b%d" % g, (n_,))).requires_grad_() for g, n_ in enumerate(widths)]
    rs = [T(pf.uniform("glin/r%d" % g, (rows, n_))) for g, n_ in enumerate(widths)]
    ys = [F.linear(x, w, b) for w, b in zip(ws, bs)]
    skip = 4   # this output gets no gradient (an unused style vector)
    sum((y * r).sum() for g, (y, r) in enumerate(zip(ys, rs)) if g != skip).backward()
    for accumulate in (False, True):
        xg = x.detach().float().to(dev()).requires_grad_()
        wg = [w.detach().float().to(dev()).requires_grad_() for w in ws]
        bg = [b.detach().float().to(dev()).requires_grad_() for b in bs]
        pre = 0.5
        if accumulate:   # existing .grad buffers (FusedAdam's flat bucket): the backward adds into them
            for t_ in wg + bg:
                t_.grad = torch.full_like(t_, pre)
        yg = ops.grouped_linear(xg, list(zip(wg, bg)))
        assert len(yg) == len(widths)
        for y_, y in zip(yg, ys):
            assert y_.is_contiguous() and relerr(y_, y) < TOL
        sum((y_ * r.float().to(dev())).sum() for g, (y_, r) in enumerate(zip(yg, rs)) if g != skip).backward()
        assert relerr(xg.grad, x.grad) < TOL
        off = pre if accumulate else 0.0
        for g in range(len(widths)):
            if g == skip:
                assert wg[g].grad is None or float((wg[g].grad - off).abs().max()) == 0.0
                continue
            assert relerr(wg[g].grad - off, ws[g].grad) < TOL, g
            assert relerr(bg[g].grad - off, bs[g].grad, atol=1e-9) < TOL, g


@pytest.mark.parametrize("shape", [(6, 4, 3, 4), (64, 32, 3, 8), (512, 512, 3, 4), (3, 64, 9, 8), (16, 128, 1, 4)])
def test_sn_conv_sequence(shape):
    """SNConv2d = spectral_norm(Conv2d): 3 training calls + 1 eval call; outputs, u/v buffers and grads
    (incl. the gradient through sigma) against the oracle's restatement of torch's hook."""
    from optimalstrategiesagainstgenerativeattacks_amd import model_blocks as mb
    Cout, Cin, K, H = shape
    tag = "sn%s" % (shape,)
    sd = {"bias": T(pf.fill_value("bias", (Cout,), tag)),
          "weight_orig": T(pf.fill_value("weight_orig", (Cout, Cin, K, K), tag)),
          "weight_u": T(pf.fill_value("weight_u", (Cout,), tag)),
          "weight_v": T(pf.fill_value("weight_v", (Cin * K * K,), tag))}
    go.set_requires_grad(sd)
    mod = mb.SNConv2d(Cin, Cout, K, padding=(K - 1) // 2)
    mod.load_state_dict({k: v.detach().float() for k, v in sd.items()})
    mod.to(dev())
    x = T(pf.normal(tag + "x", (2, Cin, H, H)))
    for i in range(4):
        training = i < 3
        mod.train(training)
        y = go.sn_conv(sd, "", x, training)
        yg = mod(nhwc(x))
        assert relerr(nchw(yg), y) < TOL, i
        assert relerr(mod.weight_u, sd["weight_u"]) < TOL, i
        assert relerr(mod.weight_v, sd["weight_v"]) < TOL, i
        if i == 1:
            r = T(pf.uniform(tag + "r", tuple(y.shape)))
            (y * r).sum().backward()
            (yg * nhwc(r)).sum().backward()
            assert relerr(mod.weight_orig.grad, sd["weight_orig"].grad) < TOL, "d weight_orig"
            assert relerr(mod.bias.grad, sd["bias"].grad) < TOL, "d bias"


@pytest.mark.parametrize("N,C,H", [(2, 8, 4), (3, 64, 16), (2, 3, 8), (4, 130, 2), (5, 512, 1), (2, 1, 8)])
def test_instance_norm(N, C, H):
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    tag = "in%d_%d_%d" % (N, C, H)
    x = T(pf.normal(tag + "x", (N, C, H, H)) * 2 + 0.5).requires_grad_()
    w = T(1 + 0.3 * pf.uniform(tag + "w", (C,))).requires_grad_()
    b = T(0.3 * pf.uniform(tag + "b", (C,))).requires_grad_()
    y = go.instance_norm(x, w, b)
    r = T(pf.uniform(tag + "r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    wg = w.detach().float().to(dev()).requires_grad_()
    bg = b.detach().float().to(dev()).requires_grad_()
    yg = ops.instance_norm(xg, wg, bg)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad, atol=1e-6) < 2e-4
    assert relerr(wg.grad, w.grad, atol=1e-6) < TOL
    assert relerr(bg.grad, b.grad) < TOL
    if H == 1:  # output is exactly the bias, dx = 0
        assert float((yg - bg.view(1, 1, 1, C)).abs().max()) == 0.0
        assert float(xg.grad.abs().max()) == 0.0


@pytest.mark.parametrize("N,C,H,res", [(2, 5, 4, False), (3, 64, 8, True), (2, 512, 2, True)])
def test_ada_in(N, C, H, res):
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    tag = "ada%d_%d_%d" % (N, C, H)
    x = T(pf.normal(tag + "x", (N, C, H, H))).requires_grad_()
    ms = T(pf.normal(tag + "m", (N, C))).requires_grad_()
    ss = T(pf.normal(tag + "s", (N, C))).requires_grad_()
    rs = T(pf.normal(tag + "res", (N, C, H, H))).requires_grad_() if res else None
    y = go.ada_in(x, ms, ss)
    if res:
        y = y + rs
    r = T(pf.uniform(tag + "r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    mg = ms.detach().float().to(dev()).requires_grad_()
    sg = ss.detach().float().to(dev()).requires_grad_()
    rg = nhwc(rs).requires_grad_() if res else None
    yg = ops.ada_in(xg, mg, sg, res=rg)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < 2e-4
    assert relerr(mg.grad, ms.grad) < TOL
    assert relerr(sg.grad, ss.grad) < TOL
    if res:
        assert relerr(nchw(rg.grad), rs.grad) < TOL


def test_pools_tanh_layout():
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    x = T(pf.normal("pool/x", (3, 70, 8, 8))).requires_grad_()
    y = F.avg_pool2d(x, 2)
    r = T(pf.uniform("pool/r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    yg = ops.avg_pool2(xg)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL

    x = T(pf.normal("mp/x", (3, 70, 4, 4))).requires_grad_()
    x.data[0, 0] = 1.25  # ties: torch routes the gradient to the first maximum
    y = go.lrelu(F.adaptive_max_pool2d(x, (1, 1)).view(3, -1))
    r = T(pf.uniform("mp/r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    yg = ops.maxpool_lrelu(xg)
    assert relerr(yg, y) < TOL
    (yg * r.float().to(dev())).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < TOL

    x = T(pf.normal("tanh/x", (2, 3, 8, 8))).requires_grad_()
    y = torch.tanh(x)
    r = T(pf.uniform("tanh/r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = x.detach().float().to(dev()).requires_grad_()
    yg = ops.tanh(xg)
    assert relerr(yg, y) < TOL
    (yg * r.float().to(dev())).sum().backward()
    assert relerr(xg.grad, x.grad) < TOL

    x = T(pf.normal("lay/x", (4, 3, 8, 8)))
    xg = x.float().to(dev()).requires_grad_()
    yg = ops.to_nhwc(xg)
    assert relerr(yg, x.permute(0, 2, 3, 1)) < 1e-7
    zg = ops.to_nchw(yg * 2.0)
    assert relerr(zg, 2 * x) < 1e-7
    zg.sum().backward()
    assert relerr(xg.grad, torch.full_like(x, 2.0)) < 1e-7


@pytest.mark.parametrize("N,C,H", [(2, 16, 4), (3, 128, 16), (2, 256, 8)])
def test_self_attention_block(N, C, H):
    from optimalstrategiesagainstgenerativeattacks_amd import model_blocks as mb
    tag = "att%d_%d_%d/" % (N, C, H)
    mod = mb.SelfAttention(C)
    sd = {k: T(pf.fill_value(k, tuple(v.shape), tag)) for k, v in mod.state_dict().items()}
    go.set_requires_grad(sd)
    mod.load_state_dict({k: v.detach().float() for k, v in sd.items()})
    mod.to(dev()).train()
    x = T(pf.normal(tag + "x", (N, C, H, H))).requires_grad_()
    y = go.self_attention(sd, "", x, True)
    r = T(pf.uniform(tag + "r", tuple(y.shape)))
    (y * r).sum().backward()
    xg = nhwc(x).requires_grad_()
    yg = mod(xg)
    assert relerr(nchw(yg), y) < TOL
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(xg.grad), x.grad) < 1e-4
    gmax = max(float(sd[k].grad.norm()) for k, _ in mod.named_parameters())
    for k, p in mod.named_parameters():
        assert relerr_floor(p.grad, sd[k].grad, 1e-3 * gmax) < 2e-4, k


def test_head_and_losses():
    from optimalstrategiesagainstgenerativeattacks_amd import gim_img_models as gm
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    from optimalstrategiesagainstgenerativeattacks_amd.gim_basic_models import GIMMeanStdFcStat
    D = 32
    for n, k in [(5, 10), (1, 1), (3, 1)]:
        tag = "head%d_%d/" % (n, k)
        dis = gm.GIMFaceDis(D, D, GIMMeanStdFcStat(D, 2, (2 * D, 3 * D, 2 * D)))
        sd = {kk: T(pf.fill_value(kk, tuple(v.shape), tag)) for kk, v in dis.state_dict().items()}
        go.set_requires_grad(sd)
        dis.load_state_dict({kk: v.detach().float() for kk, v in sd.items()})
        dis.to(dev())
        ins = {nm: T(pf.normal(tag + nm, (4, t, D))).requires_grad_() for nm, t in
               [("test_src", n), ("test_env", n), ("si_src", k), ("si_env", k)]}
        out = go.face_dis(sd, "", **ins)
        loss = go.gan_loss(out, 1.0) + go.gan_loss(out, 0.0)
        loss.mean().backward()
        gin = {nm: v.detach().float().to(dev()).requires_grad_() for nm, v in ins.items()}
        og = dis(**gin)
        assert relerr(og, out) < TOL
        lg = ops.bce_logits(og, 1.0).squeeze() + ops.bce_logits(og, 0.0).squeeze()
        assert relerr(lg, loss) < TOL
        lg.mean().backward()
        for nm in ins:
            assert relerr(gin[nm].grad, ins[nm].grad, atol=1e-8) < 1e-4, (n, k, nm)
        for kk, p in dis.named_parameters():
            assert relerr(p.grad, sd[kk].grad, atol=1e-8) < 1e-4, (n, k, kk)


def test_generator_glue():
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    B, t, D = 3, 5, 32
    env = T(pf.normal("glue/env", (B, D))).requires_grad_()
    w = T(pf.normal("glue/w", (B, t, D))).requires_grad_()
    for rm in (True, False):
        env.grad = w.grad = None
        y = env.unsqueeze(1) + (w - w.mean(1, keepdim=True) if rm else w)
        r = T(pf.uniform("glue/r", tuple(y.shape)))
        (y * r).sum().backward()
        eg = env.detach().float().to(dev()).requires_grad_()
        wg = w.detach().float().to(dev()).requires_grad_()
        yg = ops.noise_combine(eg, wg, rm)
        assert relerr(yg, y) < TOL
        (yg * r.float().to(dev())).sum().backward()
        assert relerr(eg.grad, env.grad) < TOL and relerr(wg.grad, w.grad) < TOL
    x = T(pf.normal("glue/x", (B, t, D))).requires_grad_()
    y = x.mean(1)
    (y * y).sum().backward()
    xg = x.detach().float().to(dev()).requires_grad_()
    yg = ops.mean_dim1(xg)
    (yg * yg).sum().backward()
    assert relerr(yg, y) < TOL and relerr(xg.grad, x.grad) < TOL
    s = T(pf.normal("glue/s", (B, D))).requires_grad_()
    y = s.unsqueeze(1).expand(-1, t, -1)
    r = T(pf.uniform("glue/r2", (B, t, D)))
    (y * r).sum().backward()
    sg = s.detach().float().to(dev()).requires_grad_()
    yg = ops.repeat_dim1(sg, t)
    (yg * r.float().to(dev())).sum().backward()
    assert relerr(yg, y) < 1e-7 and relerr(sg.grad, s.grad) < TOL
    a = T(pf.normal("glue/a", (B * t, 2, 4, 4))).requires_grad_()
    b = T(pf.normal("glue/b", (B, 3, 4, 4)))
    y = torch.cat((a.view(B, t, 2, 4, 4), b.unsqueeze(1).expand(-1, t, -1, -1, -1)), dim=2).view(B * t, 5, 4, 4)
    r = T(pf.uniform("glue/r3", tuple(y.shape)))
    (y * r).sum().backward()
    ag = nhwc(a).requires_grad_()
    yg = ops.concat2(ag, nhwc(b), t)
    assert relerr(nchw(yg), y) < 1e-7
    (yg * nhwc(r)).sum().backward()
    assert relerr(nchw(ag.grad), a.grad) < 1e-7


def test_fused_adam_matches_torch_adam_form():
    from optimalstrategiesagainstgenerativeattacks_amd.optim import FusedAdam
    ps = [T(pf.normal("adam/p%d" % i, s)).float() for i, s in enumerate([(7,), (4, 3, 3, 3), (5, 6), (130,)])]
    ref = [p.clone().double() for p in ps]
    gp = [torch.nn.Parameter(p.clone().to(dev())) for p in ps]
    gp[1].data = gp[1].data.contiguous(memory_format=torch.channels_last)
    opt = FusedAdam([{"params": gp[:2], "lr": 1e-2}, {"params": gp[2:], "lr": 3e-3}], lr=1e-2, betas=(0.0, 0.99))
    sd = {"g0.%d" % i: r for i, r in enumerate(ref[:2])}
    sd.update({"g1.%d" % i: r for i, r in enumerate(ref[2:])})
    oad = go.Adam(sd, [("g0.", 1e-2), ("g1.", 3e-3)], betas=(0.0, 0.99))
    for it in range(4):
        opt.zero_grad()
        for i, (k, r) in enumerate(sd.items()):
            g = T(pf.normal("adam/g%d_%d" % (it, i), tuple(r.shape)))
            r.grad = g
            gp[i].grad.add_(g.float().to(dev()))
        opt.step()
        oad.step()
        for i, (k, r) in enumerate(sd.items()):
            assert relerr(gp[i], r) < 1e-5, (it, k)
    st = opt.state_dict()
    assert len(st["state"]) == 4 and len(st["param_groups"]) == 2
    assert float(st["state"][0]["step"]) == 4.0
    assert relerr(st["state"][1]["exp_avg_sq"], oad.state["g0.1"]["v"]) < 1e-5


def test_conv_operands_beyond_2gib_are_split_over_the_batch():
    """One launch addresses an operand through 32-bit buffer offsets (2 GiB); larger batches are halved inside the C ABI
    (images are independent; the accumulating wgrad adds the halves).  2.5 GB activations, 1x1 conv."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    lib = _lib.load()
    N, S, Cin, Cout = 600, 256, 16, 16
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, S, S, Cin, device=dev(), generator=g)
    w = torch.randn(Cout, 1, 1, Cin, device=dev(), generator=g) * 0.2
    assert x.numel() * 4 > 2 ** 31
    st = torch.cuda.current_stream().cuda_stream
    sh = _lib.GimConvShape(N, S, S, Cin, Cout, 1, 0, 1.0)
    y = torch.empty(N, S, S, Cout, device=dev())
    _lib.check(lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st), "fwd")
    for sl in (slice(0, 2), slice(299, 301), slice(598, 600)):      # both halves and the seam
        ref = torch.einsum("nhwc,oc->nhwo", x[sl].double(), w.view(Cout, Cin).double())
        assert relerr(y[sl], ref) < 3e-6
    dx = torch.empty_like(x)
    _lib.check(lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh, st), "dgrad")
    for sl in (slice(0, 2), slice(299, 301), slice(598, 600)):
        ref = torch.einsum("nhwo,oc->nhwc", y[sl].double(), w.view(Cout, Cin).double())
        assert relerr(dx[sl], ref) < 3e-6
    acc = torch.zeros(Cout * Cin, device=dev())
    _lib.check(lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), acc.data_ptr(), None, sh, st), "wgrad_acc")
    ref = torch.zeros(Cout, Cin, device=dev(), dtype=torch.float64)
    for n0 in range(0, N, 50):
        ref += torch.einsum("nhwo,nhwc->oc", y[n0:n0 + 50].double(), x[n0:n0 + 50].double())
    assert relerr(acc.view(Cout, Cin), ref) < 1e-4


@pytest.mark.parametrize("N,S,Cin,Cout,slope", [(80, 32, 3, 64, 1.0), (16, 64, 6, 64, 0.2), (80, 32, 64, 3, 1.0), (5, 8, 1, 96, 0.2),
                                                (3, 16, 160, 8, 0.2), (2, 4, 8, 16, 1.0)])
def test_wgrad_1x1_narrow_side(N, S, Cin, Cout, slope):
    """Weight + bias gradient of 1x1 convolutions with <= 8 channels on one side (the image skip convs 3/6 -> 64 and 64 -> 3 of
    models/gim_img_models.py): the outer-product kernel behind gim_conv2d_wgrad_acc, both orientations, channel counts off the
    64-lane tile, pixel counts off the 512-pixel slice, with the pre-activation on x, ADDING to what the arena slot already holds."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(N, S, S, Cin, device=dev(), generator=g)
    dy = torch.randn(N, S, S, Cout, device=dev(), generator=g)
    st = torch.cuda.current_stream().cuda_stream
    sh = _lib.GimConvShape(N, S, S, Cin, Cout, 1, 0, slope)
    pre_w = torch.randn(Cout * Cin, device=dev(), generator=g)
    pre_b = torch.randn(Cout, device=dev(), generator=g)
    acc, bacc = pre_w.clone(), pre_b.clone()
    _lib.check(lib.gim_conv2d_wgrad_acc(dy.data_ptr(), x.data_ptr(), acc.data_ptr(), bacc.data_ptr(), sh, st), "wgrad_acc")
    xa = F.leaky_relu(x.double(), slope)
    ref = torch.einsum("nhwo,nhwc->oc", dy.double(), xa)
    assert relerr(acc.view(Cout, Cin).double() - pre_w.view(Cout, Cin).double(), ref) < 1e-5
    assert relerr(bacc.double() - pre_b.double(), dy.double().sum((0, 1, 2))) < 1e-5
    acc2 = torch.zeros_like(acc)
    _lib.check(lib.gim_conv2d_wgrad_acc(dy.data_ptr(), x.data_ptr(), acc2.data_ptr(), None, sh, st), "wgrad_acc, no bias")
    assert relerr(acc2.view(Cout, Cin), ref) < 1e-5


@pytest.mark.parametrize("N,S,Cin,Cout,K,J,slope", [(5, 16, 3, 64, 3, 4, 0.2), (3, 32, 6, 64, 9, 2, 0.2), (2, 8, 3, 32, 9, 8, 1.0), (4, 4, 1, 16, 3, 4, 0.2),
                                                     (2, 64, 6, 64, 9, 4, 1.0), (3, 2, 8, 48, 3, 2, 0.2)])
def test_dgrad_xfold_narrow_input(N, S, Cin, Cout, K, J, slope):
    """Gradient w.r.t. images (<= 8 input channels) on x-folded weights (gim_conv2d_xfold_weights + gim_conv2d_dgrad_xfold: J adjacent
    dx pixels as the output columns of one stride-(1, J) convolution of dy) against fp64 autograd of F.conv2d, with the 1/sigma
    scale and the LeakyReLU mask; J up to the row width (one folded pixel per row), 16- and 32-column tiles."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, S, S, Cin, device=dev(), generator=g)
    dy = torch.randn(N, S, S, Cout, device=dev(), generator=g)
    w = torch.randn(Cout, K, K, Cin, device=dev(), generator=g) / np.sqrt(K * K * Cin)     # channels-last weight memory
    sigma = torch.tensor([1.7], device=dev())
    st = torch.cuda.current_stream().cuda_stream
    wx = torch.empty(J * Cin * K * (K + J - 1) * Cout, device=dev())
    _lib.check(lib.gim_conv2d_xfold_weights(w.data_ptr(), wx.data_ptr(), Cout, Cin, K, J, st), "xfold_weights")
    sh = _lib.GimConvShape(N, S, S, Cin, Cout, K, 0, slope)
    dx = torch.full((N, S, S, Cin), float("nan"), device=dev())
    mask = x if slope != 1.0 else None
    _lib.check(lib.gim_conv2d_dgrad_xfold(dy.data_ptr(), wx.data_ptr(), sigma.data_ptr(), mask.data_ptr() if mask is not None else None,
                                          dx.data_ptr(), sh, J, st), "dgrad_xfold")
    xr = x.double().permute(0, 3, 1, 2).cpu().requires_grad_()
    wr = w.double().permute(0, 3, 1, 2).cpu()
    yr = F.conv2d(F.leaky_relu(xr, slope) if slope != 1.0 else xr, wr / 1.7, padding=(K - 1) // 2)
    (yr * dy.double().permute(0, 3, 1, 2).cpu()).sum().backward()
    assert relerr(nchw(dx), xr.grad) < TOL


@pytest.mark.parametrize("N,S,Cin,Cout,slope,target", [(5, 32, 64, 128, 0.2, 0), (9, 8, 96, 256, 0.2, 64), (33, 4, 32, 128, 1.0, 0), (3, 64, 32, 128, 0.2, 700),
                                                        (7, 16, 128, 384, 0.2, 0)])
def test_wgrad_row_resident_3x3(N, S, Cin, Cout, slope, target):
    """Weight + bias gradient of plain 3x3 convolutions on the ROW-RESIDENT kernel (tile code 20000: a workgroup owns 128 output
    channels x one tap row x 32 input channels; the x rows of 16 pixels stay in LDS for the row's three taps) against fp64 autograd of
    F.conv2d: maps of 4 ... 64 pixels per row (steps of four rows, two rows, one row, part of a row), slices that end inside an image,
    the pre-activation on x, ADDING into what the arena slot holds."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    import ctypes
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(31)
    x = torch.randn(N, S, S, Cin, device=dev(), generator=g)
    dy = torch.randn(N, S, S, Cout, device=dev(), generator=g)
    st = torch.cuda.current_stream().cuda_stream
    sh = _lib.GimConvShape(N, S, S, Cin, Cout, 3, 0, slope, 0, 0, 0, 20000, 0, target)
    plan = (ctypes.c_int32 * 8)()
    lib.gim_conv_launch_plan(sh, 3, ctypes.cast(plan, ctypes.c_void_p))
    assert plan[7] == 1 and plan[4] == 3 * (Cin // 32) and plan[5] == Cout // 128, list(plan)
    pre_w = torch.randn(Cout * 9 * Cin, device=dev(), generator=g)
    pre_b = torch.randn(Cout, device=dev(), generator=g)
    acc, bacc = pre_w.clone(), pre_b.clone()
    _lib.check(lib.gim_conv2d_wgrad_acc(dy.data_ptr(), x.data_ptr(), acc.data_ptr(), bacc.data_ptr(), sh, st), "wgrad_acc")
    xr = x.double().permute(0, 3, 1, 2)
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device=dev(), requires_grad=True)
    br = torch.zeros(Cout, dtype=torch.float64, device=dev(), requires_grad=True)
    yr = F.conv2d(F.leaky_relu(xr, slope) if slope != 1.0 else xr, wr, br, padding=1)
    (yr * dy.double().permute(0, 3, 1, 2)).sum().backward()
    got = (acc - pre_w).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert relerr(got.double(), wr.grad) < 1e-5
    assert relerr((bacc - pre_b).double(), br.grad) < 1e-5


def test_deterministic_wgrad_slabs_switch():
    """GIM_WGRAD_SLABS=1 (read at import: a child process): the non-queued weight-gradient path combines its pixel slices as slabs
    + a fixed-order reduce instead of float atomics - the parity cases still pass, and two runs of one weight gradient are bit-equal."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from optimalstrategiesagainstgenerativeattacks_amd import ops\n"
        "assert ops._WGRAD_SLABS\n"
        "g = torch.Generator(device='cuda').manual_seed(1)\n"
        "x = torch.randn(8, 32, 32, 32, device='cuda', generator=g)\n"
        "w = (torch.randn(64, 32, 3, 3, device='cuda', generator=g) * 0.05).contiguous(memory_format=torch.channels_last)\n"
        "r = torch.randn(8, 32, 32, 64, device='cuda', generator=g)\n"
        "outs = []\n"
        "for _ in range(2):\n"
        "    wg = w.clone().requires_grad_()\n"
        "    (ops.conv2d(x, wg, None, None, None, None, None, 0, 0.2) * r).sum().backward()\n"
        "    outs.append(wg.grad.clone())\n"
        "assert torch.equal(outs[0], outs[1]), 'slab combine is not run-to-run identical'\n"
        "wd = w.double().requires_grad_()\n"
        "(torch.nn.functional.conv2d(torch.nn.functional.leaky_relu(x.permute(0, 3, 1, 2).double(), 0.2), wd, padding=1) * r.permute(0, 3, 1, 2).double()).sum().backward()\n"
        "err = float((outs[0].double() - wd.grad).norm() / wd.grad.norm())\n"
        "assert err < 3e-5, err\n"
        "print('ok', err)\n") % root
    r_ = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GIM_WGRAD_SLABS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        text=True, timeout=300)
    assert r_.returncode == 0 and "ok" in r_.stdout, r_.stderr[-2000:]


def test_failed_backward_leaves_no_stale_weight_gradient_jobs():
    """An exception inside autograd after some convs queued their weight-gradient jobs must not leak into the next iteration:
    FusedAdam.zero_grad() drops the jobs, re-zeroes the arena slots and re-arms the end-of-backward callback."""
    from optimalstrategiesagainstgenerativeattacks_amd import model_blocks as mb, ops
    from optimalstrategiesagainstgenerativeattacks_amd.optim import FusedAdam

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    torch.manual_seed(0)
    conv = mb.SNConv2d(16, 16, 3, padding=1).to(dev())
    opt = FusedAdam(conv.parameters(), lr=1e-3)
    x = torch.randn(2, 8, 8, 16, device=dev())

    def grads(fail):
        opt.zero_grad()
        xin = x.clone().requires_grad_()
        y = conv(Boom.apply(xin) if fail else xin)
        y.square().sum().backward()
        return conv.weight_orig.grad.clone()
    state = {k_: v.clone() for k_, v in conv.state_dict().items()}
    ref = grads(False)
    conv.load_state_dict(state)   # same u, v for the repeat
    with pytest.raises(RuntimeError, match="boom"):
        grads(True)               # the conv's wgrad job is queued, then the graph raises before the flush
    assert ops.wgrad_queue.jobs
    conv.load_state_dict(state)
    again = grads(False)
    assert not ops.wgrad_queue.jobs
    assert relerr(again, ref) < 1e-5


def test_integration_md_ctypes_stub_runs():
    """The reference-side ctypes binding shown in INTEGRATION.md is executed as written and compared with the package's op."""
    import os
    import re
    from optimalstrategiesagainstgenerativeattacks_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = next(b for b in blocks if "def conv2d_nhwc" in b)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)   # the stub opens the library by its repository-relative path
    try:
        exec(stub, ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(3, 8, 8, 32, device=dev(), generator=g)
    w = torch.randn(16, 32, 3, 3, device=dev(), generator=g).contiguous(memory_format=torch.channels_last) * 0.1
    b = torch.randn(16, device=dev(), generator=g)
    sigma = torch.tensor([1.7], device=dev())
    y_stub = ns["conv2d_nhwc"](x, w.permute(0, 2, 3, 1).contiguous(), b, sigma, leaky_slope=0.2)
    y_pkg = ops.conv2d(x, w, b, None, sigma, None, None, 0, 0.2)
    assert torch.equal(y_stub, y_pkg)
