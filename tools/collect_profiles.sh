#!/bin/bash
# Copy the judged summaries of one tools/measure_round.sh run from gpurun_out/measure_<tag>/ into profiles/<tag>_*:  tools/collect_profiles.sh <tag>
tag=$1
src=gpurun_out/measure_$tag
for f in $src/bench_*.log; do cp $f profiles/${tag}_$(basename $f); done
cp $src/conv_shapes_fp32.txt profiles/${tag}_conv_shapes_vox64_B16_fp32.txt
cp $src/stats/r_kernel_stats.csv profiles/${tag}_kernel_stats_bench_vox64_B16.csv
python tools/pmc_summary.py $src > profiles/${tag}_pmc_single_kernels.txt
python tools/dominant_kernel_from_trace.py $src/stats/r_kernel_trace.csv > profiles/${tag}_dominant_kernel_in_step.txt 2>/dev/null || rm -f profiles/${tag}_dominant_kernel_in_step.txt
grep -E "passed|failed" $src/pytest_gpu.log | tail -1 > profiles/${tag}_pytest_gpu_summary.txt
ls profiles | grep "^${tag}_"
