"""Read a rocprofv3 --kernel-trace CSV of a SINGLE-STREAM run and print, for the last repetition of the step, where the chain's time
goes: kernel time by kernel name, idle gaps between consecutive kernels, the longest individual launches.
    python tools/trace_chain.py <kernel_trace.csv> [n_repeats_in_trace=10]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:70]) for r in rows))
# the trace holds warm-up iterations and `reps` repetitions of the same launch sequence: take the last 1/reps of the timed part by count
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
assert len(adam) >= 2, "no adam_kernel launches in the trace"
lo, hi = adam[-2] + 1, adam[-1] + 1    # one step = from after the previous Adam launch to this one
step = ev[lo:hi]
t0, t1 = step[0][0], step[-1][1]
busy = sum(e[1] - e[0] for e in step)
gaps = [(step[i + 1][0] - step[i][1], step[i][2], step[i + 1][2]) for i in range(len(step) - 1)]
idle = sum(max(g[0], 0) for g in gaps)
print("last step: %d launches, wall %.2f ms, kernel time %.2f ms, idle between kernels %.2f ms (mean gap %.1f us)"
      % (len(step), (t1 - t0) / 1e6, busy / 1e6, idle / 1e6, idle / 1e3 / max(len(gaps), 1)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, nm in step:
    by[nm][0] += 1
    by[nm][1] += e - s
print("kernel time by name:")
for nm, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print("   %-72s %4d x %8.1f us = %7.3f ms" % (nm, c, t / c / 1e3, t / 1e6))
print("largest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print("   %7.1f us after %-50s before %s" % (g / 1e3, a[:50], b[:50]))
hist = collections.Counter(min(int(max(g[0], 0) / 1e3), 20) for g in gaps)
print("gap histogram (us: count):", dict(sorted(hist.items())))
