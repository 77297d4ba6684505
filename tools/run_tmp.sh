#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02am; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc: $(grep -E 'passed|failed' $out/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
[ $rc -ge 124 ] && exit $rc
Q="--no-cpu-baseline --no-kernel-bench --no-traffic --no-bf16x3"
for i in 1 2 3; do
  timeout -k 10 300 python bench.py $Q > $out/bench_on_$i.log 2>&1; echo "act storage on  $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_on_$i.log | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' $out/bench_on_$i.log)"
  GIM_NO_ACT_STORAGE=1 timeout -k 10 300 python bench.py $Q > $out/bench_off_$i.log 2>&1; echo "act storage off $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_off_$i.log | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' $out/bench_off_$i.log)"
done
