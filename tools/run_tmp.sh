#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02ag; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc: $(grep -E 'passed|failed' $out/pytest_gpu.log | tail -1)"; grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head
[ $rc -ge 124 ] && exit $rc
Q="--no-cpu-baseline --no-kernel-bench --no-traffic --no-bf16x3"
for i in 1 2; do timeout -k 10 300 python bench.py $Q > $out/bench_$i.log 2>&1; echo "bench $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$i.log | head -1)"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16x3 --no-traffic --no-kernel-bench > $GRAFT_REPO_ROOT/$out/stats.log 2>&1; echo "stats rc=$?"
grep -h "bgemm" $GRAFT_REPO_ROOT/$out/stats/r_kernel_stats.csv | cut -c1-120
