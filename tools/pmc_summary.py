"""Condense the PMC passes of tools/measure_round.sh (gpurun_out/measure_<tag>/pmc/<probe>/<set>/r_counter_collection.csv) into one
table: per probe the mean counter values over the conv kernel's launches.    python tools/pmc_summary.py gpurun_out/measure_r02f"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
print("rocprofv3 --kernel-trace --pmc <set> -- python3 tools/kernel_probe.py <kind> N H Cin Cout K ups reps pool   (tools/measure_round.sh; one counter set per run)")
print("values = mean over the conv kernel's launches; SQ_* cycle counters are in quad-cycles, summed over the chip\n")
for probe in sorted(os.listdir(os.path.join(root, "pmc"))):
    d = os.path.join(root, "pmc", probe)
    if not os.path.isdir(d):
        continue
    vals, kname = collections.defaultdict(list), None
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("void conv_igemm_kernel", "void conv_igemm_patch", "void conv_wgrad_kernel", "void conv_wgrad_row_kernel")):
                kname = r["Kernel_Name"].split("(")[0].replace("void ", "")
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    k = probe.split("_")
    print("%s N=%s H=%s %s->%s K=%s pool=%s   [%s]" % (k[0], k[1], k[2], k[3], k[4], k[5], k[8], kname))
    for name in sorted(vals):
        v = vals[name]
        print("   %-28s %.4g" % (name, sum(v) / len(v)))
    if "SQ_WAVE_CYCLES" in vals:
        wc = sum(vals["SQ_WAVE_CYCLES"]) / len(vals["SQ_WAVE_CYCLES"])
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if nm in vals:
                print("   %-28s %.1f %% of SQ_WAVE_CYCLES" % (nm, 100 * sum(vals[nm]) / len(vals[nm]) / wc))
    print()
