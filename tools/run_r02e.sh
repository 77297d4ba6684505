#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02e; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_tuned_rows.py > $out/pytest_main.log 2>&1; rc=$?
echo "pytest main rc=$rc"; grep -E "passed|failed" $out/pytest_main.log | tail -2; grep -E "^FAILED|^ERROR" $out/pytest_main.log | head
[ $rc -ge 124 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_tuned_rows.py -m gpu -q > $out/pytest_tuned.log 2>&1; rc=$?; echo "tuned rows rc=$rc"; tail -2 $out/pytest_tuned.log; grep -E "^FAILED" $out/pytest_tuned.log | head -5
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1; echo "shapes rc=$?"; tail -1 $out/conv_shapes_fp32.txt
timeout -k 10 400 python bench.py --no-cpu-baseline --no-traffic > $out/bench.log 2>&1; echo "bench rc=$?: $(grep -o '"value": [0-9.]*, "unit"' $out/bench.log | head -1) $(grep -o '"bf16x3_path": {"value": [0-9.]*' $out/bench.log) $(grep -o '"executed_frac": [0-9.]*' $out/bench.log)"
