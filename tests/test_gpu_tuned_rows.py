"""Parity of the launch configurations the BENCHMARKS run.

``csrc/conv_tune_table.inc`` (written by tools/conv_autotune.py on an MI355X) holds, per layer shape of the bench
workloads, the output-tile shape, the split-K factor and the wgrad slice target that beat the heuristic.  Its rows are keyed
on the kernel-level problem size (M = images x pixels, channels, K), so the small-batch parity tests elsewhere never hit
them.  This file walks EVERY distinct convolution shape of the table and runs forward, input gradient and weight / bias
gradient through the product's autograd operator (``ops.conv2d``: the same C-ABI calls, shape struct and queued
weight-gradient path as a training step; the launch-override fields of the shape stay 0, so the table row is what
launches), and compares ELEMENTWISE with torch's own fp64 ``F.conv2d`` autograd of the unfused reference form
``avg_pool2d(conv2d(upsample(leaky_relu(x)), w) + b)``  (models/model_blocks.py:497-514, 752-773, 842-865).

The CPU part (``-m "not gpu"``) checks with ``gim_conv_launch_plan`` that every row of the table is reachable: the shape in
its comment makes the launcher find exactly that row.
"""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "optimalstrategiesagainstgenerativeattacks_amd", "csrc", "conv_tune_table.inc")
ROW = re.compile(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},\s*// (\w+) ([\d.,]+):")
TOL = 3e-5          # relative L2 over the whole tensor, fp32 kernels against fp64 (same as tests/test_gpu_ops.py)
TOL_MAX = 2e-4      # largest single-element error relative to the largest reference element


def parse_table():
    """[(kind, M, Ca, Cb, Ktot, pc, tile, ks, kind name, cfg)] with cfg = (N, H, W, Cin, Cout, K, ups, slope, pool, fold)."""
    rows = []
    for line in open(TABLE):
        mt = ROW.search(line)
        if not mt:
            assert line.lstrip().startswith("//") or not line.strip(), "unparsed table line: %r" % line
            continue
        v = mt.group(10).split(",")
        cfg = tuple(int(t) for t in v[:7]) + (float(v[7]), int(v[8]), int(v[9]))
        rows.append(tuple(int(mt.group(i)) for i in range(1, 9)) + (mt.group(9), cfg))
    return rows


ROWS = parse_table()
SHAPES = sorted({r[9] for r in ROWS})


def _plan(lib, cfg, api_kind):
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    N, H, W, Cin, Cout, K, ups, slope, pool, fold = cfg
    sh = _lib.GimConvShape(N, H, W, Cin, Cout, K, ups, slope, pool, fold, 0)
    out = (ctypes.c_int32 * 8)()
    assert lib.gim_conv_launch_plan(sh, api_kind, ctypes.cast(out, ctypes.c_void_p)) == 0, lib.gim_last_error()
    return list(out)


def test_table_parses_and_is_not_empty():
    assert len(ROWS) > 100 and len(SHAPES) > 40
    assert {r[0] for r in ROWS} <= {0, 1, 2, 4}


def test_every_table_row_is_reachable_from_its_shape():
    """The launcher, given the layer shape in a row's comment, finds that row (table kind -> entry point / matrix path:
    0 fwd, 1 dgrad on k-major weights, 2 wgrad, 4 dgrad on transposed weights) and
    launches its tile / split.  No GPU needed: gim_conv_launch_plan launches nothing."""
    from optimalstrategiesagainstgenerativeattacks_amd import _lib
    lib = _lib.load()
    api = {0: 0, 1: 1, 2: 3, 4: 2}   # table kind -> plan kind (entry point)
    tiles = {128: (128, 128), 641: (64, 128), 1264: (128, 64), 64: (64, 64), 6432: (64, 64)}
    for kind, M, Ca, Cb, Ktot, pc, tile, ks, name, cfg in ROWS:
        plan = _plan(lib, cfg, api[kind])
        assert plan[0] == 1, ("row not found by its own shape", kind, name, cfg, plan)
        if kind == 2:
            continue   # wgrad rows carry a workgroup target, not a tile
        if tile:   # tile code + 20000: the patch-resident kernel (plain 3x3 layers), reported in plan[7]
            assert tuple(plan[1:3]) == tiles[tile % 20000], (kind, name, cfg, plan, tile)
            assert plan[7] & 255 == (1 if tile >= 20000 else 0), (kind, name, cfg, plan, tile)   # (bits 8..: skipped share of the K steps)
        # the split-K factor is the row's, capped by the number of K steps and rounded to whole steps per slice
        assert 1 <= plan[3] <= max(ks, 1), (kind, name, cfg, plan, ks)


# ------------------------------------------------------------------------------------------------------------------
# GPU: elementwise parity of every tuned shape
# ------------------------------------------------------------------------------------------------------------------
_REF_DEV = {}


def _ref_device():
    """Where the fp64 reference runs: on the GPU through torch's own double-precision convolution (ATen's im2col + rocBLAS
    dgemm: nothing of this repo) when that agrees with the CPU result on a probe, else on the CPU."""
    if "dev" not in _REF_DEV:
        dev = torch.device("cpu")
        try:
            g = torch.Generator().manual_seed(1)
            x = torch.randn(2, 5, 8, 8, dtype=torch.float64, generator=g)
            w = torch.randn(7, 5, 3, 3, dtype=torch.float64, generator=g)
            outs = []
            for d in ("cpu", "cuda:0"):
                xx, ww = x.detach().clone().to(d).requires_grad_(), w.detach().clone().to(d).requires_grad_()   # fresh leaves per device
                y = F.avg_pool2d(F.conv2d(F.leaky_relu(xx, 0.2), ww, padding=1), 2)
                y.square().sum().backward()
                outs.append([t.detach().cpu() for t in (y, xx.grad, ww.grad)])
            if all(relerr(a, b) < 1e-12 for a, b in zip(outs[1], outs[0])):
                dev = torch.device("cuda:0")
        except Exception:   # noqa: BLE001 - any failure of the fp64 GPU path: use the CPU
            dev = torch.device("cpu")
        _REF_DEV["dev"] = dev
    return _REF_DEV["dev"]


def _reference(x, w, b, dy, cfg, rdev):
    """fp64 F.conv2d autograd of the unfused form; NHWC fp32 cuda tensors in, (y, dx, dw, db) fp64 out (NHWC / [Cout,K,K,Cin])."""
    N, H, W, Cin, Cout, K, ups, slope, pool, fold = cfg
    xr = x.to(rdev, torch.float64).permute(0, 3, 1, 2).contiguous().requires_grad_()
    wr = w.to(rdev, torch.float64).permute(0, 3, 1, 2).contiguous().requires_grad_()     # [Cout,Cin,K,K]
    br = b.to(rdev, torch.float64).requires_grad_()
    xa = F.leaky_relu(xr, slope) if slope != 1.0 else xr
    if ups:
        xa = F.interpolate(xa, scale_factor=2, mode="nearest")
    y = F.conv2d(xa, wr, br, padding=(K - 1) // 2)
    if pool:
        y = F.avg_pool2d(y, 2)
    y.backward(dy.to(rdev, torch.float64).permute(0, 3, 1, 2).contiguous())
    return (y.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.permute(0, 2, 3, 1), br.grad)


def _close(got, ref, what):
    ref = ref.to(got.device)
    err = float((got.double() - ref).norm() / ref.norm())
    emax = float((got.double() - ref).abs().max() / ref.abs().max())
    assert err < TOL and emax < TOL_MAX, "%s: relative L2 %.2e (max element %.2e)" % (what, err, emax)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", SHAPES, ids=[",".join(str(c) for c in s) for s in SHAPES])
def test_tuned_shape_elementwise_vs_fp64_conv2d(cfg):
    from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    lib = _lib.load()
    N, H, W, Cin, Cout, K, ups, slope, pool, fold = cfg
    kinds = {r[0] for r in ROWS if r[9] == cfg}
    g = torch.Generator(device=dev).manual_seed(hash(cfg) & 0xFFFF)
    Hs, Ws = H >> ups, W >> ups
    x = torch.randn(N, Hs, Ws, Cin, device=dev, generator=g)
    w = torch.randn(Cout, K, K, Cin, device=dev, generator=g) / (Cin * K * K) ** 0.5     # [Cout][K][K][Cin] storage
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.rand(N, H >> pool, W >> pool, Cout, device=dev, generator=g) * 2 - 1
    y_r, dx_r, dw_r, db_r = _reference(x, w, b, dy, cfg, _ref_device())
    # the launches below must be the table's: ask the launcher (same shape struct the operator builds)
    api = {0: 0, 1: 1, 2: 3, 4: 2}      # table kind -> entry point of gim_conv_launch_plan
    for k in kinds & set(api):
        assert _plan(lib, cfg, api[k])[0] == 1, ("table row not in force", k, cfg)
    xg = x.clone().requires_grad_()
    wg = w.permute(0, 3, 1, 2).detach().requires_grad_()          # logical [Cout,Cin,K,K], channels-last storage
    assert wg.permute(0, 2, 3, 1).is_contiguous()
    bg = b.clone().requires_grad_()
    # .grad buffers in the weights' own memory order, as FusedAdam's flat gradient bucket provides them: the backward
    # then takes the queued path of a training step (gim_conv2d_wgrad_acc into an arena + batched finish)
    wg.grad = torch.zeros_like(wg)
    bg.grad = torch.zeros_like(bg)
    # sigma = 1 with u = v = 0: the spectral-norm chain rule of the finish runs (as for every conv of the engine) and
    # adds nothing, so the reference stays the plain convolution
    sg, u0, v0 = torch.ones(1, device=dev), torch.zeros(Cout, device=dev), torch.zeros(Cin * K * K, device=dev)
    yg = ops.conv2d(xg, wg, bg, None, sg, u0, v0, ups, slope, pool=bool(pool))
    _close(yg, y_r, "forward")
    yg.backward(dy)
    torch.cuda.synchronize()
    _close(xg.grad, dx_r, "input gradient")
    _close(wg.grad.permute(0, 2, 3, 1), dw_r, "weight gradient")
    _close(bg.grad, db_r, "bias gradient")
