"""Concurrency picture of the training step from a rocprofv3 kernel trace (--kernel-trace --output-format csv):
how much of the wall time has 0 / 1 / 2+ kernels in flight, which kernel classes fill it, the idle gaps, and per-queue busy time.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-bench
    python tools/trace_timeline.py OUT/**/r_kernel_trace.csv [first_step last_step]
"""
import collections
import csv
import glob
import sys


def klass(name):
    if "conv_igemm" in name:
        return "conv fwd/dgrad"
    if "conv_wgrad" in name:
        return "conv wgrad"
    if "adam" in name:
        return "adam"
    if "norm_" in name:
        return "norm"
    if "snb_" in name or "sn_" in name:
        return "spectral"
    if "wgq_" in name:
        return "wgrad finish"
    if "bgemm" in name or "softmax" in name or "attn" in name:
        return "attention"
    if "at::native" in name or "rocclr" in name:
        return "torch / copies"
    return "other pointwise"


def main():
    path = glob.glob(sys.argv[1], recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[0])
    # steps: split at the Adam kernels of the discriminator (2 adam launches per step)
    adam = [e for e in ev if "adam_kernel" in e[2]]
    n_steps = len(adam) // 2
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else n_steps // 2
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else n_steps - 1
    t0, t1 = adam[2 * lo][0], adam[2 * hi][0]
    sel = [e for e in ev if e[0] >= t0 and e[0] < t1]
    steps = hi - lo
    wall = (t1 - t0) / 1e6 / steps
    print("%s: steps %d..%d of %d, %.3f ms per step, %d launches per step" % (path, lo, hi, n_steps, wall, len(sel) // steps))
    # sweep
    pts = []
    for s, e, nm, q in sel:
        pts.append((s, 1, klass(nm)))
        pts.append((min(e, t1), -1, klass(nm)))
    pts.sort()
    live = collections.Counter()
    depth_time = collections.Counter()
    combo_time = collections.Counter()
    last = t0
    nlive = 0
    for t, d, k in pts:
        dt = t - last
        if dt > 0:
            depth_time[min(nlive, 4)] += dt
            combo_time[tuple(sorted(kk for kk, c in live.items() if c > 0))] += dt
        live[k] += d
        nlive += d
        last = t
    tot = sum(depth_time.values())
    print("kernels in flight:  " + "  ".join("%s: %.1f %%" % ("4+" if d == 4 else d, 100.0 * v / tot) for d, v in sorted(depth_time.items())))
    print("what is running (share of wall time, top 14):")
    for combo, v in combo_time.most_common(14):
        print("   %5.1f %%  %s" % (100.0 * v / tot, " + ".join(combo) if combo else "(idle)"))
    by_class = collections.Counter()
    for s, e, nm, q in sel:
        by_class[klass(nm)] += e - s
    print("kernel time per step by class (sum of durations, overlapping): " + ", ".join("%s %.2f ms" % (k, v / 1e6 / steps) for k, v in by_class.most_common()))
    by_q = collections.Counter()
    for s, e, nm, q in sel:
        by_q[q] += e - s
    print("busy time per queue per step: " + ", ".join("q%s %.2f ms" % (q, v / 1e6 / steps) for q, v in sorted(by_q.items())))


if __name__ == "__main__":
    main()
