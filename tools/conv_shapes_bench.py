"""Per-layer roofline table: record every convolution / linear shape one training iteration launches, then time
the forward, dgrad and wgrad kernels of each distinct shape through the C ABI with HIP events.

    python tools/conv_shapes_bench.py [--workload vox64] [--batch 16] > gpurun_out/conv_shapes.txt
"""
import argparse
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops  # noqa: E402


def time_ms(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vox64")
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    u = bench.UNIT[args.workload]
    (m, n, k), B = u.get("mnk", (1, 5, 10)), args.batch
    G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
    trainer = G.DataParallelMock(tr)
    leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1234)

    fwd_calls, bwd_calls = collections.Counter(), collections.Counter()
    orig_f, orig_b = ops.ConvFn.forward, ops.ConvFn.backward

    def rec_f(ctx, *a):
        y = orig_f(ctx, *a)
        fwd_calls[ctx.cfg[:8] + (int(ctx.cfg[10]), int(ctx.cfg[11]))] += 1
        return y

    def rec_b(ctx, dy):
        bwd_calls[(ctx.cfg[:8] + (int(ctx.cfg[10]), int(ctx.cfg[11])), bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]))] += 1
        return orig_b(ctx, dy)

    ops.ConvFn.forward, ops.ConvFn.backward = staticmethod(rec_f), staticmethod(rec_b)
    G.gim_step(trainer, leaked, real, si)
    torch.cuda.synchronize()
    ops.ConvFn.forward, ops.ConvFn.backward = staticmethod(orig_f), staticmethod(orig_b)

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    tot = collections.Counter()
    for cfg, cnt in fwd_calls.items():
        N, H, W, Cin, Cout, KH, ups, slope, pool, fold = cfg
        n_dx = sum(c for (cf, dx, dw), c in bwd_calls.items() if cf == cfg and dx)
        n_dw = sum(c for (cf, dx, dw), c in bwd_calls.items() if cf == cfg and dw)
        sh = _lib.GimConvShape(N, H, W, Cin, Cout, KH, ups, slope, pool, fold, 0)
        x = torch.randn(N, H >> ups, W >> ups, Cin, device=dev)
        KF = KH + 1 if fold else KH
        w = torch.randn(Cout, KF, KF, Cin, device=dev) * 0.05   # folded layout when fold
        y = torch.randn(N, H >> pool, W >> pool, Cout, device=dev)
        lowdx = 1 if (ups and fold) else 0
        dx = torch.empty(N, H >> lowdx, W >> lowdx, Cin, device=dev)
        ns = 1   # pixel slices combined with float atomics, as in the training step
        slabs = torch.zeros(ns * Cout * KF * KF * Cin, device=dev)
        flops = ops.conv_executed_flops(N, H, W, Cin, Cout, KH, ups, pool, fold)      # what the kernels execute (folds!)
        algo = ops.conv_algorithmic_flops(N, H, W, Cin, Cout, KH)                      # the unfused reference op
        rows_form = ops._ROWS_FORM and KH >= 3 and Cin <= 8 and KH * Cin <= 64 and Cout >= 16 and Cout % 4 == 0 and not (ups or pool)
        if rows_form:   # as ops.ConvFn: padded + activated image copy, row-contiguous K (the copy is part of the forward's time)
            pad = (KH - 1) // 2
            CaP = (KH * Cin + 15) & ~15
            xp = torch.empty(N, H + 2 * pad, W + 2 * pad, Cin, device=dev)
            wr = torch.empty(Cout * KH * CaP, device=dev)
            lib.gim_conv2d_pack_rows_weights(w.data_ptr(), wr.data_ptr(), Cout, Cin, KH, st)
            shr = _lib.GimConvShape(N, H, W, Cin, Cout, KH, 0, 1.0, 0, 0, 0, 0, 1)

            def fwd_rows():
                lib.gim_pad_image(x.data_ptr(), xp.data_ptr(), N, H, W, Cin, pad, slope, st)
                lib.gim_conv2d_fwd_rows(xp.data_ptr(), wr.data_ptr(), None, None, None, y.data_ptr(), shr, st)
            t_f = time_ms(fwd_rows)
        elif ops._MERGED_SUBPIXEL and ups and fold and KH >= 5 and (KH & 3) == 1 and Cout <= 4:
            # as ops.ConvFn: the four parity classes stacked into one plain convolution + the depth-to-space copy
            T = (KH + 1) // 2
            wm = torch.empty(4 * Cout * T * T * Cin, device=dev)
            lib.gim_conv2d_pack_subpixel_weights(w.data_ptr(), wm.data_ptr(), Cout, Cin, KH, st)
            y4 = torch.empty(N, H >> 1, W >> 1, 4 * Cout, device=dev)
            shm = _lib.GimConvShape(N, H >> 1, W >> 1, Cin, 4 * Cout, T, 0, slope, 0, 0, 0)

            def fwd_stacked():
                lib.gim_conv2d_fwd(x.data_ptr(), wm.data_ptr(), None, None, None, y4.data_ptr(), shm, st)
                lib.gim_depth_to_space2(y4.data_ptr(), None, y.data_ptr(), N, H >> 1, W >> 1, Cout, 1.0, st)
            t_f = time_ms(fwd_stacked)
        else:
            t_f = time_ms(lambda: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh, st))
        J = ops._xfold_factor(Cin, W) if (KH >= 3 and not (ups or pool or fold) and Cout % 16 == 0 and ops._NARROW_XFOLD) else 0
        if J:   # as ops._conv_dgrad: x-folded gradient w.r.t. images
            wx = torch.empty(J * Cin * KH * (KH + J - 1) * Cout, device=dev)
            lib.gim_conv2d_xfold_weights(w.data_ptr(), wx.data_ptr(), Cout, Cin, KH, J, st)
            t_d = time_ms(lambda: lib.gim_conv2d_dgrad_xfold(y.data_ptr(), wx.data_ptr(), None, None, dx.data_ptr(), sh, J, st)) if n_dx else 0.0
        elif Cout % 16 == 0 and not (ups and not fold) and Cin <= 8:
            wt = torch.empty(Cin * KF * KF * Cout, device=dev)
            lib.gim_conv2d_transpose_weights(w.data_ptr(), wt.data_ptr(), Cout, Cin, KF, st)
            t_d = time_ms(lambda: lib.gim_conv2d_dgrad_t(y.data_ptr(), wt.data_ptr(), None, None, dx.data_ptr(), sh, st)) if n_dx else 0.0
        else:
            t_d = time_ms(lambda: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh, st)) if n_dx else 0.0

        slabs_r = torch.zeros(Cout * KH * ((KH * Cin + 15) & ~15), device=dev) if rows_form else None

        def wg():   # the step's form: accumulate into an arena slot (the finish of all convs is two batched launches per pass)
            if rows_form:
                lib.gim_conv2d_wgrad_rows_acc(y.data_ptr(), xp.data_ptr(), slabs_r.data_ptr(), None, shr, st)
            else:
                lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), slabs.data_ptr(), None, sh, st)
        t_w = time_ms(wg) if n_dw else 0.0
        # share of the K steps the forward / dgrad kernel really runs (position-major rows on small maps skip padding taps)
        plan = (ctypes.c_int32 * 8)()
        lib.gim_conv_launch_plan(sh, 0, ctypes.cast(plan, ctypes.c_void_p))
        sf = 1.0 - (plan[7] >> 8) / 1000.0
        lib.gim_conv_launch_plan(sh, 1, ctypes.cast(plan, ctypes.c_void_p))
        sd = 1.0 - (plan[7] >> 8) / 1000.0
        total = cnt * t_f + n_dx * t_d + n_dw * t_w
        tot["fwd"] += cnt * t_f
        tot["dgrad"] += n_dx * t_d
        tot["wgrad"] += n_dw * t_w
        tot["gflop"] += flops * (cnt * sf + n_dx * sd + n_dw) / 1e9
        tot["algo"] += algo * (cnt + n_dx + n_dw) / 1e9
        rows.append((total, cfg, cnt, n_dx, n_dw, flops, t_f, t_d, t_w, ns, sf, sd))
    rows.sort(reverse=True)
    print("%-46s %4s %4s %4s %8s | %8s %6s | %8s %6s | %8s %6s %4s | %8s" %
          ("N,H,W,Cin,Cout,K,ups,slope,pool,fold", "fwd", "dx", "dw", "exe GF", "fwd ms", "TF", "dgrad ms", "TF", "wgrad ms", "TF", "S", "tot ms"))
    print("(GF = FLOPs one launch EXECUTES: the pool / sub-pixel folds run (K+1)^2 taps at a quarter of the pixels; TF = executed TFLOP/s)")
    for total, cfg, cnt, n_dx, n_dw, flops, t_f, t_d, t_w, ns, sf, sd in rows:
        tf = lambda t, sh_=1.0: flops * sh_ / t / 1e9 if t else 0.0  # noqa: E731
        print("%-46s %4d %4d %4d %8.2f | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f %4d | %8.2f%s" %
              (",".join(str(c) for c in cfg), cnt, n_dx, n_dw, flops / 1e9, t_f, tf(t_f, sf), t_d, tf(t_d, sd), t_w, tf(t_w), ns, total,
               "   (padding taps skipped: fwd runs %.0f %%, dgrad %.0f %% of its K steps)" % (100 * sf, 100 * sd) if min(sf, sd) < 1.0 else ""))
    s = tot["fwd"] + tot["dgrad"] + tot["wgrad"]
    print("sum of conv/linear kernels per step: %.1f ms (fwd %.1f, dgrad %.1f, wgrad %.1f); executed %.0f GFLOP -> %.1f TFLOP/s = %.3f of the "
          "fp32 MFMA peak (algorithmic, unfused ops at full resolution: %.0f GFLOP -> %.1f TFLOP/s)"
          % (s, tot["fwd"], tot["dgrad"], tot["wgrad"], tot["gflop"], tot["gflop"] / s, tot["gflop"] / s / 157.3, tot["algo"], tot["algo"] / s))


if __name__ == "__main__":
    main()
