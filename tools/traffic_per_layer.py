"""Per-layer L2-miss traffic from tools/traffic_per_layer.sh: FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE per launch of the
conv kernel against the bytes of the operands touched once.    python tools/traffic_per_layer.py gpurun_out/<outdir>"""
import csv
import glob
import os
import sys

root = sys.argv[1]


CONV = ("void conv_igemm_kernel", "void conv_igemm_patch", "void conv_wgrad_kernel", "void conv_wgrad_row_kernel")
EXTRA = ("pad_image_kernel",)      # the image copy of the row-contiguous form: part of every forward launch


def mean_kib(d):
    """Counter value per conv launch: the conv kernel's own, plus the per-launch helper kernels of its launch form."""
    total, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(CONV):
                total += float(r["Counter_Value"])
                n += 1
            elif r["Kernel_Name"].startswith(EXTRA):
                total += float(r["Counter_Value"])
    return total / n if n else float("nan")


print("%-6s %-28s %9s %9s %9s | %9s %7s   (MB per launch; operands = x + w + y touched once)" % ("kind", "N,H,Cin,Cout,K,pool", "fetch x2", "write", "total", "operands", "ratio"))
tot_t = tot_o = 0.0
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d).split("_")
    kind, (N, H, Cin, Cout, K, ups, reps, pool) = name[0], [int(v) for v in name[1:9]]
    fetch, write = mean_kib(os.path.join(d, "FETCH_SIZE")) * 1024 * 2 / 1e6, mean_kib(os.path.join(d, "WRITE_SIZE")) * 1024 / 1e6
    KF = K + 1 if pool else K
    xb, yb, wb = N * H * H * Cin * 4 / 1e6, N * (H >> pool) ** 2 * Cout * 4 / 1e6, Cout * KF * KF * Cin * 4 / 1e6
    ops = xb + yb + wb
    print("%-6s %-28s %9.1f %9.1f %9.1f | %9.1f %7.2f" % (kind, "%d,%d,%d,%d,%d,%d" % (N, H, Cin, Cout, K, pool), fetch, write, fetch + write, ops, (fetch + write) / ops))
    tot_t += fetch + write
    tot_o += ops
print("sum over these launches: %.0f MB against %.0f MB of operands: %.2fx" % (tot_t, tot_o, tot_t / tot_o))
