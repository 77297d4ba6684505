#!/bin/bash
# round-2 GPU session c: failing tests re-run, capture probes, A/B of the zero pool / narrow dgrad, wgrad autotune
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02c; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_tuned_rows.py > $out/pytest_main.log 2>&1; rc=$?
echo "pytest main rc=$rc"; grep -E "passed|failed" $out/pytest_main.log | tail -2; grep -E "^FAILED|^ERROR" $out/pytest_main.log | head
[ $rc -ge 124 ] && exit $rc
for c in t_fork t_originwait t_fork2 t_selfwait t_alias fwd bwd full gimstep; do
  timeout -k 10 120 python tools/capture_probe.py $c > $out/cap_$c.log 2>&1; rc=$?
  echo "== capture $c rc=$rc: $(grep -E '^\s+\[|^case|Fatal|File' $out/cap_$c.log | tail -4 | tr '\n' '|' | cut -c1-400)"
  [ $rc -eq 124 ] && exit 124
done
B="--no-cpu-baseline --no-kernel-bench --no-traffic --no-bf16x3 --steps 20 --warmup 5"
for v in base GIM_NO_ZERO_POOL GIM_NO_NARROW_DGRAD_T; do
  if [ $v = base ]; then timeout -k 10 300 python bench.py $B > $out/bench_$v.log 2>&1; else env $v=1 timeout -k 10 300 python bench.py $B > $out/bench_$v.log 2>&1; fi
  rc=$?; echo "bench $v rc=$rc: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$v.log | head -1) $(grep -o '"ms_per_step_median": [0-9.]*' $out/bench_$v.log)"
  [ $rc -ge 124 ] && exit $rc
done
timeout -k 10 900 python tools/conv_autotune.py --kinds wgrad --append --write > $out/autotune_wgrad_vox64_B16.txt 2>&1; rc=$?
echo "autotune rc=$rc"; tail -3 $out/autotune_wgrad_vox64_B16.txt; [ $rc -ge 124 ] && exit $rc
cp optimalstrategiesagainstgenerativeattacks_amd/csrc/conv_tune_table.inc $out/conv_tune_table.inc
make -C optimalstrategiesagainstgenerativeattacks_amd/csrc -j8 > $out/make.log 2>&1 || { echo make failed; exit 1; }
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1; echo "shapes rc=$?"; tail -1 $out/conv_shapes_fp32.txt
timeout -k 10 300 python bench.py $B > $out/bench_tuned.log 2>&1; echo "bench tuned: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_tuned.log | head -1)"
timeout -k 10 600 python -m pytest tests/test_gpu_tuned_rows.py -m gpu -q -k "1,0,0 or 1,1.0" > $out/pytest_tuned_subset.log 2>&1; echo "tuned subset rc=$?"; tail -2 $out/pytest_tuned_subset.log
