"""HBM bytes per step from the two PMC passes of tools/measure_round.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs),
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2: 128-byte requests tallied at 64 B;
WRITE_SIZE exact; both in KiB).  Kernels of the model construction are excluded by name (at::native fills / copies before the
first conv launch are part of every run and are small).

    python tools/hbm_traffic.py gpurun_out/measure_r1k 3 vox64 16 > profiles/r01_k_hbm_traffic_vox64_B16.json
"""
import csv
import glob
import json
import sys


def total_kib(d):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    per_kernel = {}
    tot = 0.0
    for r in csv.DictReader(open(f)):
        v = float(r["Counter_Value"])
        tot += v
        k = r["Kernel_Name"].split("(")[0][:60]
        per_kernel[k] = per_kernel.get(k, 0.0) + v
    return tot, per_kernel


def main():
    root, steps, workload, B = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    fetch, fk = total_kib(root + "/pmc_fetch")
    write, wk = total_kib(root + "/pmc_write")
    fb, wb = fetch * 1024 * 2 / steps, write * 1024 / steps
    top = sorted(fk.items(), key=lambda kv: -kv[1])[:8]
    print(json.dumps({
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-kernel-bench  (tools/measure_round.sh)",
        "workload": "%s B=%d" % (workload, B), "steps_in_run": steps,
        "fetch_size_kb_raw": fetch, "write_size_kb_raw": write,
        "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B), WRITE_SIZE exact; includes model construction kernels, "
                      "divided by all %d steps run" % steps,
        "hbm_bytes_per_step": fb + wb, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
        "top_fetch_kernels_gb_per_step": {k: round(v * 1024 * 2 / steps / 1e9, 2) for k, v in top},
    }, indent=1))


if __name__ == "__main__":
    main()
