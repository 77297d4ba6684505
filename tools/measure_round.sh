#!/bin/bash
# End-of-iteration measurement set on the GPU box (one gpurun call):  tools/measure_round.sh <tag>
#   full GPU suite, default bench (traffic + CPU baseline), other workloads, rocprofv3 kernel stats, per-layer table, PMC of the
#   dominant conv kernels.  Everything lands in gpurun_out/measure_<tag>/.
# the HIP runtime reads this when it starts - under rocprofv3 --pmc the profiler initialises the GPU before python imports the package, so set it here
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/measure_$tag
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc: $(grep -E 'passed|failed' $out/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
[ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python bench.py > $out/bench_vox64_B16.log 2> $out/bench_vox64_B16.err || exit 1
echo "bench: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_vox64_B16.log | head -1) $(grep -o '"executed_frac": [0-9.]*' $out/bench_vox64_B16.log)"
Q="--no-cpu-baseline --no-kernel-bench --no-traffic"
timeout -k 10 300 python bench.py --workload om32 $Q > $out/bench_om32_B32.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --batch 64 $Q > $out/bench_vox64_B64.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --reg-param 10 $Q > $out/bench_vox64_B16_r1.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --workload vox128 --matrix-path fp16 $Q > $out/bench_vox128_B2_fp16.log 2>&1 || exit 1   # the opt-in 16-bit operand path (BASELINE config 5)
GIM_NO_STEP_OVERLAP=1 timeout -k 10 300 python bench.py $Q > $out/bench_vox64_B16_eager_sequential.log 2>&1 || exit 1   # what the graph replays, launched eagerly
timeout -k 10 300 python bench.py --workload vox128 $Q > $out/bench_vox128_B2.log 2>&1 || exit 1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 GIM_FORCE_ALLREDUCE=1 timeout -k 10 300 python bench.py $Q > $out/bench_vox64_B16_rccl1rank.log 2>&1 || exit 1
timeout -k 10 300 python bench.py $Q > $out/bench_vox64_B16_again.log 2>&1 || exit 1   # box drift check: the default again, after the sustained load above
for f in om32_B32 vox64_B64 vox64_B16_r1 vox128_B2_fp16 vox64_B16_eager_sequential vox128_B2 vox64_B16_rccl1rank vox64_B16_again; do echo "$f: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$f.log | head -1)"; done
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1 || exit 1
tail -1 $out/conv_shapes_fp32.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-bench > $out/stats.log 2>&1 || exit 1
# PMC of the dominant conv shapes (one counter set per run), forward / dgrad / wgrad with the geometry the step launches
mkdir -p $out/pmc
for probe in "fwd 80 32 64 128 3 0 10 0" "dgrad 80 32 64 128 3 0 10 0" "wgrad 80 32 64 128 3 0 10 0" "fwd 160 64 64 64 3 0 10 1" "dgrad 160 64 64 64 3 0 10 1" "wgrad 160 64 64 64 3 0 10 1" "wgrad 80 8 512 512 3 0 10 1"; do
  name=$(echo $probe | tr ' ' '_')
  for set in "MfmaUtil" "MeanOccupancyPerActiveCU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    sname=$(echo $set | cut -d' ' -f1)
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc/$name/$sname -o r -- python3 $R/tools/kernel_probe.py $probe > $out/pmc/${name}_$sname.log 2>&1 || { echo "pmc $name $sname failed"; break; }
  done
done
echo measured
