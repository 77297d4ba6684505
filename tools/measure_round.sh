#!/bin/bash
# End-of-iteration measurement set on the GPU box (one gpurun call):  tools/measure_round.sh <tag>
#   bench (default, om32, vox64 B=64), rocprofv3 kernel stats of the default bench, HBM traffic PMC passes.
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/measure_$tag
mkdir -p $out
cd $R
python bench.py > $out/bench_vox64_B16.log 2>&1 || exit 1
python bench.py --workload om32 --no-cpu-baseline --no-kernel-bench > $out/bench_om32_B32.log 2>&1 || exit 1
python bench.py --batch 64 --no-cpu-baseline --no-kernel-bench > $out/bench_vox64_B64.log 2>&1 || exit 1
python bench.py --reg-param 10 --no-cpu-baseline --no-kernel-bench > $out/bench_vox64_B16_r1.log 2>&1 || exit 1
python bench.py --graph --no-cpu-baseline --no-kernel-bench > $out/bench_vox64_B16_graph.log 2>&1 || exit 1
python bench.py --workload vox128 --no-cpu-baseline --no-kernel-bench > $out/bench_vox128_B2.log 2>&1 || exit 1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 GIM_FORCE_ALLREDUCE=1 python bench.py --no-cpu-baseline --no-kernel-bench > $out/bench_vox64_B16_rccl1rank.log 2>&1 || exit 1
python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1 || exit 1
GIM_CONV_PREC=1 python tools/conv_shapes_bench.py > $out/conv_shapes_bf16x3.txt 2>&1 || exit 1
python tools/host_overhead.py > $out/host_overhead.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16x3 > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-bench --no-bf16x3 > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-bench --no-bf16x3 > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x3 -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-bench > $out/stats_x3.log 2>&1 || exit 1
echo measured
