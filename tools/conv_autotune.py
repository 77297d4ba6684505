"""Per-layer launch autotuner: record every convolution / linear shape of one training iteration, time each kernel kind
(fwd, dgrad, wgrad) under every tile configuration x split-K factor (wgrad: slice targets) through the C ABI, and write
optimalstrategiesagainstgenerativeattacks_amd/csrc/conv_tune_table.inc with the winners that beat the heuristic by > 3 %.

    python tools/conv_autotune.py [--workload vox64] [--batch 16] [--write]      (then rebuild the library)
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import _lib, ops  # noqa: E402


def time_ms(fn, reps=12):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def record_shapes(workload, B):
    dev = torch.device("cuda:0")
    u = bench.UNIT[workload]
    m, n, k = u.get("mnk", (1, 5, 10))
    G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
    trainer = G.DataParallelMock(tr)
    leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1234)
    fwd, bwd = collections.Counter(), collections.Counter()
    orig_f, orig_b = ops.ConvFn.forward, ops.ConvFn.backward

    def rec_f(ctx, *a):
        y = orig_f(ctx, *a)
        fwd[ctx.cfg[:8] + (int(ctx.cfg[10]), int(ctx.cfg[11]))] += 1
        return y

    def rec_b(ctx, dy):
        bwd[(ctx.cfg[:8] + (int(ctx.cfg[10]), int(ctx.cfg[11])), bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]))] += 1
        return orig_b(ctx, dy)
    ops.ConvFn.forward, ops.ConvFn.backward = staticmethod(rec_f), staticmethod(rec_b)
    G.gim_step(trainer, leaked, real, si, overlap=False)
    torch.cuda.synchronize()
    ops.ConvFn.forward, ops.ConvFn.backward = staticmethod(orig_f), staticmethod(orig_b)
    return fwd, bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vox64")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--append", action="store_true", help="keep the entries already in the table (other workloads)")
    ap.add_argument("--min-gain", type=float, default=0.03)
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad", help="which kernel kinds to tune (comma separated)")
    ap.add_argument("--out", default=None, help="with --write: write the rows to this file instead of the compiled-in table")
    ap.add_argument("--only-n", type=int, default=0, help="tune only the shapes with this many images")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    fwd, bwd = record_shapes(args.workload, args.batch)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    entries, report = [], []
    tot_auto = tot_best = 0.0
    for cfg, cnt in sorted(fwd.items(), key=lambda kv: -kv[1]):
        N, H, W, Cin, Cout, KH, ups, slope, pool, fold = cfg
        if args.only_n and N != args.only_n:
            continue
        narrow = bool(Cin % 16 or Cout % 16)   # generic-K layers (3 / 6 channels): forward / dgrad use the heuristics, wgrad is tuned
        n_dx = sum(c for (cf, dx, dw), c in bwd.items() if cf == cfg and dx)
        n_dw = sum(c for (cf, dx, dw), c in bwd.items() if cf == cfg and dw)

        def shape(tile=-1, ks=0, tg=0):   # tile -1: heuristics only (ignore the rows already in the compiled-in table)
            return _lib.GimConvShape(N, H, W, Cin, Cout, KH, ups, slope, pool, fold, 0, tile, ks, tg)
        sh = shape()
        x = torch.randn(N, H >> ups, W >> ups, Cin, device=dev)
        KF = KH + 1 if fold else KH
        w = torch.randn(Cout, KF, KF, Cin, device=dev) * 0.05
        y = torch.randn(N, H >> pool, W >> pool, Cout, device=dev)
        lowdx = 1 if (ups and fold) else 0
        dx = torch.empty(N, H >> lowdx, W >> lowdx, Cin, device=dev)
        acc = torch.zeros(Cout * KF * KF * Cin, device=dev)
        up_fold = bool(ups and fold)
        # kernel-level problem sizes (what the table is keyed on)
        Mf = N * (H >> pool) * (W >> pool) if not up_fold else N * (H >> 1) * (W >> 1)
        Tf = (KF * KF if pool else (((KH + 1) // 2) ** 2 if up_fold else KH * KH))
        keys = {"fwd": (0, Mf, Cin, Cout, Tf * Cin, 1 if up_fold else 0)}
        if pool:
            Md, Td, pcd = N * (H >> 1) * (W >> 1), ((KH + 1) // 2) ** 2, 1
        elif up_fold:
            Md, Td, pcd = N * (H >> 1) * (W >> 1), KF * KF, 0
        else:
            Md, Td, pcd = N * H * W, KH * KH, 0
        keys["dgrad"] = (1, Md, Cout, Cin, Td * Cout, pcd)
        rows = Cin if up_fold else Cout
        cols = KF * KF * (Cout if up_fold else Cin)
        Mw = N * (H >> (1 if fold else 0)) * (W >> (1 if fold else 0))
        keys["wgrad"] = (2, Mw, rows, cols, KH, (1 if pool else 0) + (2 if up_fold else 0))
        dgrad_fn = lambda sh_: lib.gim_conv2d_dgrad(y.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), sh_, st)   # noqa: E731
        runs = {"fwd": (cnt, lambda sh_: lib.gim_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), sh_, st)),
                "dgrad": (n_dx, dgrad_fn),
                "wgrad": (n_dw, lambda sh_: lib.gim_conv2d_wgrad_acc(y.data_ptr(), x.data_ptr(), acc.data_ptr(), None, sh_, st))}
        for kind, (calls, fn) in runs.items():
            if not calls or kind not in args.kinds.split(",") or (narrow and kind != "wgrad"):
                continue
            Cb = keys[kind][3]
            t_auto = time_ms(lambda: fn(sh))
            best = (t_auto, 0, 0)
            if kind == "wgrad":   # output tile x workgroup target of the pixel slicing (an explicit target lifts the 512-pixel floor)
                cands = [(tl, 0, tg) for tl in (0, 128, 641, 1264, 64, 6432) for tg in (128, 256, 512, 1024, 2048, 4096, 8192)]
            else:
                tiles = [128, 641, 1264, 64] if Cb > 64 else ([1264, 64] if Cb > 32 else [])
                if tiles and keys[kind][2] % 32 == 0:
                    tiles.append(6432)   # 64x64 tile with a 32-deep K step
                cands = [(tl, ks, 0) for tl in tiles for ks in (1, 2, 3, 4, 6, 8)]
            for tl, ks, tg in cands:
                shc = shape(tl if tl else -1, ks, tg)
                t = time_ms(lambda: fn(shc), reps=10)
                if t < best[0]:
                    best = (t, tl, ks if kind != "wgrad" else tg)
            tot_auto += calls * t_auto
            tot_best += calls * best[0]
            gain = t_auto / best[0] - 1.0
            report.append((calls * (t_auto - best[0]), kind, cfg, calls, t_auto, best))
            if best[1] or best[2]:
                if gain > args.min_gain:
                    kd, M_, Ca_, Cb_, Kt_, pc_ = keys[kind]
                    entries.append((kd, M_, Ca_, Cb_, Kt_, pc_, best[1], best[2], gain, kind, cfg))
    report.sort(reverse=True)
    print("conv kernels per step: heuristic %.2f ms, best-per-layer %.2f ms" % (tot_auto, tot_best))
    for sav, kind, cfg, calls, t_auto, best in report[:40]:
        print("%-6s %-44s x%-3d auto %.3f ms  best %.3f ms (tile %d, ks/target %d)  saves %.3f ms/step"
              % (kind, ",".join(str(c) for c in cfg), calls, t_auto, best[0], best[1], best[2], sav))
    if args.write:
        path = args.out or os.path.join(ROOT, "optimalstrategiesagainstgenerativeattacks_amd", "csrc", "conv_tune_table.inc")
        old, seen = "", set()
        if args.append and os.path.exists(path):
            old = open(path).read()
            import re
            for mm in re.finditer(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \d+, \d+\}", old):
                seen.add(tuple(int(v) for v in mm.groups()))
        with open(path, "w") as f:
            f.write(old)
            f.write("// generated by tools/conv_autotune.py --workload %s --batch %d on an MI355X; entries beat the heuristic by > %d %%\n"
                    % (args.workload, args.batch, int(args.min_gain * 100)))
            for kd, M_, Ca_, Cb_, Kt_, pc_, tl, ks, gain, kind, cfg in entries:
                key = (kd, M_, Ca_, Cb_, Kt_, pc_)
                if key in seen:
                    continue
                seen.add(key)
                f.write("    {%d, %d, %d, %d, %d, %d, %d, %d},  // %s %s: +%.0f %%\n" % (kd, M_, Ca_, Cb_, Kt_, pc_, tl, ks, kind,
                                                                                     ",".join(str(c) for c in cfg), gain * 100))
        print("wrote %s (%d entries)" % (path, len(seen)))


if __name__ == "__main__":
    main()
