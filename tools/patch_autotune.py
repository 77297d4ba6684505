"""Patch-resident kernel (conv_igemm_patch_kernel) against the tap-major kernel, per plain 3x3 layer of one training step: the table /
heuristic launch of the tap-major loop and every (tile, split-K) of the patch-resident one (tile code + 20000), through the C ABI.

    python tools/patch_autotune.py [--workload vox64] [--batch 16] [--out gpurun_out/patch_rows.inc]
(prints the rows in the format of csrc/conv_tune_table.inc for the layers where a patch-resident launch wins by > 2 %, and for the
 others the row that keeps the tap-major kernel)"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimalstrategiesagainstgenerativeattacks_amd import _lib  # noqa: E402
from tools.conv_autotune import record_shapes, time_ms  # noqa: E402

CODE = {(128, 128): 128, (64, 128): 641, (128, 64): 1264, (64, 64): 64}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vox64")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--out", default=None)
    ap.add_argument("--kinds", default="fwd,dgrad,pool,wgrad", help="fwd, dgrad (plain 3x3), pool (pool-folded forward), wgrad (plain 3x3)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    fwd, bwd = record_shapes(args.workload, args.batch)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    lines, tot_a, tot_b = [], 0.0, 0.0
    for cfg, cnt in sorted(fwd.items(), key=lambda kv: -kv[1]):
        N, H, W, Cin, Cout, KH, ups, slope, pool, fold = cfg
        # (the pool-folded forward's patch-resident kernel was removed in round 4: profiles/r03_l_patch_resident_pool_fold.txt)
        if KH != 3 or ups or pool or fold or Cin % 16 or Cout % 16 or H * W < 64:
            continue
        n_dx = sum(c for (cf, dx, dw), c in bwd.items() if cf == cfg and dx)
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        y = torch.randn(N, H, W, Cout, device=dev)
        dx = torch.empty(N, H, W, Cin, device=dev)

        def shape(tile=0, ks=0):
            return _lib.GimConvShape(N, H, W, Cin, Cout, 3, 0, slope, 0, 0, 0, tile, ks, 0)
        n_dw = sum(c for (cf, dx_, dw), c in bwd.items() if cf == cfg and dw)
        if n_dw and Cin % 32 == 0 and Cout % 128 == 0 and "wgrad" in args.kinds:   # row-resident weight gradient against the column-tile kernel
            accw = torch.zeros(Cout * 9 * Cin, device=dev)

            def wshape(tile=0, tg=0):
                return _lib.GimConvShape(N, H, W, Cin, Cout, 3, 0, slope, 0, 0, 0, tile, 0, tg)
            wfn = lambda s_: lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), accw.data_ptr(), None, s_, st)   # noqa: E731
            t_old = min(time_ms(lambda: wfn(wshape())) for _ in range(3))
            bestw = (1e9, 0)
            for tg in (256, 512, 768, 1024, 1536, 2048, 3072, 4096):
                tw = time_ms(lambda: wfn(wshape(20000, tg)), reps=10)
                if tw < bestw[0]:
                    bestw = (tw, tg)
            t_new = min(bestw[0], time_ms(lambda: wfn(wshape(20000, bestw[1]))))
            usew = t_new < t_old * 0.98
            tot_a += n_dw * t_old
            tot_b += n_dw * (t_new if usew else t_old)
            print("wgrad %-30s x%-2d column-tile kernel (table row) %.3f ms | row-resident best (target %d) %.3f ms  %+.1f %%"
                  % (",".join(str(c) for c in cfg[:6]), n_dw, t_old, bestw[1], t_new, 100 * (t_old / t_new - 1)), flush=True)
            if usew:
                lines.append("    {2, %d, %d, %d, 3, 0, 20000, %d},  // wgrad %s: row-resident %.3f ms (column-tile kernel %.3f, row-resident %.3f)"
                             % (N * H * W, Cout, 9 * Cin, bestw[1], ",".join(str(c) for c in cfg), t_new, t_old, t_new))
        runs = {"fwd": (cnt, 0, lambda s_: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), s_, st), Cout, (0, N * H * W, Cin, Cout, 9 * Cin, 0)),
                "dgrad": (n_dx, 1, lambda s_: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), s_, st), Cin, (1, N * H * W, Cout, Cin, 9 * Cout, 0))}
        for kind, (calls, pk, fn, Cb, key) in runs.items():
            if not calls or kind not in args.kinds:
                continue
            plan = (ctypes.c_int32 * 8)()
            lib.gim_conv_launch_plan(shape(), pk, ctypes.cast(plan, ctypes.c_void_p))
            tcode, tks = CODE.get((plan[1], plan[2]), 64), plan[3]
            t_tap = min(time_ms(lambda: fn(shape(tcode, tks))) for _ in range(3))
            best = (1e9, 0, 0)
            tiles = [128, 641, 1264, 64] if Cb > 64 else [1264, 64]
            for tl in tiles:
                for ks in (1, 2, 3, 4, 6, 8):
                    t = time_ms(lambda: fn(shape(tl + 20000, ks)), reps=10)
                    if t < best[0]:
                        best = (t, tl, ks)
            t_patch = min(best[0], time_ms(lambda: fn(shape(best[1] + 20000, best[2]))))
            use_patch = t_patch < t_tap * 0.98
            tot_a += calls * t_tap
            tot_b += calls * min(t_tap, t_patch if use_patch else t_tap)
            print("%-5s %-30s x%-2d tap-major (tile %d, ks %d) %.3f ms | patch-resident best (tile %d, ks %d) %.3f ms  %+.1f %%"
                  % (kind, ",".join(str(c) for c in cfg[:6]), calls, tcode, tks, t_tap, best[1], best[2], t_patch, 100 * (t_tap / t_patch - 1)), flush=True)
            if use_patch:   # the tile the launcher really took (128-row tiles fall back to 64 rows on maps below 128 pixels or wider than 64)
                lib.gim_conv_launch_plan(shape(best[1] + 20000, best[2]), pk, ctypes.cast(plan, ctypes.c_void_p))
                best = (best[0], CODE[(plan[1], plan[2])], plan[3])
            tl, ks = (best[1] + 20000, best[2]) if use_patch else (tcode, tks)
            lines.append("    {%d, %d, %d, %d, %d, %d, %d, %d},  // %s %s: %s %.3f ms (tap-major %.3f, patch-resident %.3f)"
                         % (key + (tl, ks, kind, ",".join(str(c) for c in cfg), "patch-resident" if use_patch else "tap-major", min(t_tap, t_patch), t_tap, t_patch)))
    print("plain 3x3 fwd + dgrad and pool-fold fwd per step: tap-major %.2f ms -> %.2f ms with the better kernel per layer" % (tot_a, tot_b))
    print("\n".join(lines))
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
