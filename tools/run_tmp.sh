#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02ah; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc: $(grep -E 'passed|failed' $out/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1; tail -1 $out/conv_shapes_fp32.txt
Q="--no-cpu-baseline --no-kernel-bench --no-traffic --no-bf16x3"
for i in 1 2; do timeout -k 10 300 python bench.py $Q > $out/bench_$i.log 2>&1; echo "bench $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$i.log | head -1)"; done
