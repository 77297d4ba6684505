"""Launch tuning against the time of the WHOLE training step.

tools/conv_autotune.py picks, per layer shape, the tile / split that is fastest for that kernel ALONE.  Inside the step two lanes
of kernels overlap (gim_step): the ramp-up and the tail of one kernel are filled by the other lane, so a configuration with the
better steady state and the worse tail can be the better one there - and the reverse.  This tool tunes the rows that carry the
most FLOPs by coordinate descent on the measured step time itself: for one (kind, shape) at a time every candidate is forced
through ops._TUNE_OVERRIDE (the per-call launch-override fields of gim_conv_shape), the step is timed, and a candidate is kept
only if it beats the incumbent twice by more than the noise margin.

    python tools/step_autotune.py [--workload vox64] [--batch 16] [--top 14] [--steps 12] [--out gpurun_out/step_tune.inc]
    (prints the rows in the format of csrc/conv_tune_table.inc; merge them with tools/merge_tune_rows.py)
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from optimalstrategiesagainstgenerativeattacks_amd import ops  # noqa: E402


def table_key(kind, cfg):
    """(table kind, M, Ca, Cb, Ktot | KH, pc) of a layer shape, as csrc/conv_igemm.hip keys its launch table (tools/conv_autotune.py)."""
    N, H, W, Cin, Cout, KH, ups, pool, fold = cfg
    KF = KH + 1 if fold else KH
    up_fold = bool(ups and fold)
    if kind == "fwd":
        Mf = N * (H >> pool) * (W >> pool) if not up_fold else N * (H >> 1) * (W >> 1)
        Tf = (KF * KF if pool else (((KH + 1) // 2) ** 2 if up_fold else KH * KH))
        return (0, Mf, Cin, Cout, Tf * Cin, 1 if up_fold else 0)
    if kind == "dgrad":
        if pool:
            Md, Td, pcd = N * (H >> 1) * (W >> 1), ((KH + 1) // 2) ** 2, 1
        elif up_fold:
            Md, Td, pcd = N * (H >> 1) * (W >> 1), KF * KF, 0
        else:
            Md, Td, pcd = N * H * W, KH * KH, 0
        return (1, Md, Cout, Cin, Td * Cout, pcd)
    rows = Cin if up_fold else Cout
    cols = KF * KF * (Cout if up_fold else Cin)
    Mw = N * (H >> (1 if fold else 0)) * (W >> (1 if fold else 0))
    return (2, Mw, rows, cols, KH, (1 if pool else 0) + (2 if up_fold else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vox64")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--top", type=int, default=14, help="rows per kind, by executed FLOPs per step")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--margin", type=float, default=0.004, help="relative gain a candidate must show (twice) to be kept")
    ap.add_argument("--budget", type=float, default=600.0, help="seconds")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    u = bench.UNIT[args.workload]
    (m, n, k), B = u.get("mnk", (1, 5, 10)), args.batch
    G, tr = bench.build_trainer(u["S"], u["C"], n, m, k, dev)
    trainer = G.EpisodeParallel(tr)
    leaked, real, si = bench.synthetic_batch(B, m, n, k, u["C"], u["S"], dev, 1234)
    zgen = torch.Generator(device=dev).manual_seed(4321)

    def step():
        z = torch.randn((B, n, 512), device=dev, generator=zgen)
        tr.do_global_step()
        return G.gim_step(trainer, leaked, real, si, z=z, defer_join=True)

    def measure(steps=args.steps):
        for _ in range(2):
            step()
        ops.join_lanes()
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream()
        e0 = cur.record_event(torch.cuda.Event(enable_timing=True))
        marks = []
        for _ in range(steps):
            step()
            marks.append(cur.record_event(torch.cuda.Event(enable_timing=True)))
        ops.join_lanes()
        torch.cuda.synchronize()
        ts = [e0.elapsed_time(mk) for mk in marks]
        per = [b - a for a, b in zip([0.0] + ts[:-1], ts)]
        return statistics.median(per)

    for _ in range(3):
        step()
    with ops.count_flops() as counts:
        step()
        ops.join_lanes()
        torch.cuda.synchronize()
    rows = {"fwd": [], "dgrad": [], "wgrad": []}
    for (kind, cfg), (nl, f) in counts.items():
        if kind in rows and cfg[3] % 16 == 0 and cfg[4] % 16 == 0:
            rows[kind].append((nl * f, cfg, nl))
    order = []
    for kind in rows:
        rows[kind].sort(reverse=True)
        order += [(fl, kind, cfg, nl) for fl, cfg, nl in rows[kind][:args.top]]
    order.sort(reverse=True)

    def set_override(kind, cfg, ov):
        if ov is None:
            ops._TUNE_OVERRIDE.pop((kind, cfg), None)
        else:
            ops._TUNE_OVERRIDE[(kind, cfg)] = ov
        ops._SPLITS_K.clear()

    base = statistics.median([measure() for _ in range(3)])
    print("baseline step: %.3f ms (median of 3 x %d steps)" % (base, args.steps), flush=True)
    t_start = time.time()
    kept = []
    best = base
    for fl, kind, cfg, nl in order:
        if time.time() - t_start > args.budget:
            print("time budget reached", flush=True)
            break
        N, H, W, Cin, Cout, KH, ups, pool, fold = cfg
        Cb = Cin if kind == "dgrad" else Cout
        if kind == "wgrad":
            cands = [(tl, tg) for tl in (128, 641, 1264, 64, 6432) for tg in (512, 1024, 2048)]
        else:
            tiles = [128, 641, 1264, 64, 6432] if Cb > 64 else [1264, 64, 6432]
            cands = [(tl, 1) for tl in tiles]
        incumbent = measure()
        results = []
        for ov in cands:
            set_override(kind, cfg, ov)
            results.append((measure(), ov))
        set_override(kind, cfg, None)
        results.sort()
        t_best, ov_best = results[0]
        line = "%-6s %-36s x%-2d %7.1f GF: table/heuristic %.3f ms | best %s %.3f ms" % (kind, ",".join(str(c) for c in cfg), nl, fl / 1e9, incumbent, ov_best, t_best)
        if t_best < incumbent * (1 - args.margin):
            # confirm: incumbent and candidate again, interleaved
            a1 = measure()
            set_override(kind, cfg, ov_best)
            b1 = measure()
            set_override(kind, cfg, None)
            a2 = measure()
            set_override(kind, cfg, ov_best)
            b2 = measure()
            if max(b1, b2) < min(a1, a2) * (1 - args.margin / 2):
                kept.append((kind, cfg, ov_best, min(a1, a2), max(b1, b2)))
                best = max(b1, b2)
                line += "  KEPT (confirm %.3f / %.3f vs %.3f / %.3f)" % (b1, b2, a1, a2)
            else:
                set_override(kind, cfg, None)
                line += "  not confirmed (%.3f / %.3f vs %.3f / %.3f)" % (b1, b2, a1, a2)
        print(line, flush=True)
    final = statistics.median([measure() for _ in range(3)])
    print("step with the kept overrides: %.3f ms (baseline %.3f ms): %+.2f %%" % (final, base, 100 * (final / base - 1)))
    lines = []
    for kind, cfg, ov, ta, tb in kept:
        kd, M_, Ca_, Cb_, Kt_, pc_ = table_key(kind, cfg)
        N, H, W, Cin, Cout, KH, ups, pool, fold = cfg
        name = "%d,%d,%d,%d,%d,%d,%d,%s,%d,%d" % (N, H, W, Cin, Cout, KH, ups, "1.0" if KH == 1 and not pool else "0.2", pool, fold)
        lines.append("    {%d, %d, %d, %d, %d, %d, %d, %d},  // %s %s: step %.3f -> %.3f ms" % (kd, M_, Ca_, Cb_, Kt_, pc_, ov[0], ov[1], kind, name, ta, tb))
    print("\n".join(lines))
    if args.out:
        with open(args.out, "w") as f:
            f.write("// generated by tools/step_autotune.py --workload %s --batch %d on an MI355X (rows chosen by the time of the whole step)\n" % (args.workload, args.batch))
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
