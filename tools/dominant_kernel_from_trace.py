"""Durations of the dominant convolution's launches INSIDE the training step, from the rocprofv3 --kernel-trace of the bench command
(rocprofv3 --stats aggregates by kernel template, which mixes layer shapes; the dispatch's grid identifies the shape).
    python tools/dominant_kernel_from_trace.py <r_kernel_trace.csv> [images=80] [H=32] [Cin=64] [Cout=128]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
N, H, Cin, Cout = (int(a) for a in (sys.argv[2:6] + ["80", "32", "64", "128"][len(sys.argv) - 2:]))
M = N * H * H
want = {"fwd (A = x, 64x128 tile)": ("conv_igemm_kernel<64, 128, 1, 2, 0, 0, 16, 0>", (M // 64 * 256, (Cout + 127) // 128, 1)),
        "dgrad (A = dy, k-major weights)": ("conv_igemm_kernel<", (None, None, None))}
print("shape: %d images %dx%d, %d -> %d channels, 3x3 (M = %d output pixels)" % (N, H, H, Cin, Cout, M))
by = {}
for r in rows:
    nm = r["Kernel_Name"]
    if not nm.startswith(("void conv_igemm_kernel", "void conv_igemm_patch_kernel", "void conv_wgrad_kernel")):
        continue
    g = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    by.setdefault((nm.split("(")[0].replace("void ", ""), g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# forward: M / BM workgroups along x with BM in {64, 128}, Cout / BN along y; dgrad: the same M with Cin output channels
for (nm, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    wg_x = g[0] // 256
    if nm.startswith("conv_igemm") and wg_x in (M // 64, M // 128) and g[2] == 1:
        v.sort()
        print("   %-52s grid %-18s %4d launches: mean %7.1f us, median %7.1f us, min %7.1f us" % (nm, g, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
