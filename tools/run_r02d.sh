#!/bin/bash
# round-2 GPU session d: suite, wgrad autotune for the other workloads and the bf16x3 path, kernel stats
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02d; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_tuned_rows.py > $out/pytest_main.log 2>&1; rc=$?
echo "pytest main rc=$rc"; grep -E "passed|failed" $out/pytest_main.log | tail -2; grep -E "^FAILED|^ERROR" $out/pytest_main.log | head
[ $rc -ge 124 ] && exit $rc
T=optimalstrategiesagainstgenerativeattacks_amd/csrc/conv_tune_table.inc
for wl in "om32 32" "vox64 64" "vox128 2"; do
  set -- $wl
  timeout -k 10 600 python tools/conv_autotune.py --workload $1 --batch $2 --kinds wgrad --append --write > $out/autotune_wgrad_$1_B$2.txt 2>&1; rc=$?
  echo "autotune $1 B$2 rc=$rc: $(grep 'conv kernels per step' $out/autotune_wgrad_$1_B$2.txt)"; [ $rc -ge 124 ] && exit $rc
done
GIM_CONV_PREC=1 timeout -k 10 600 python tools/conv_autotune.py --kinds wgrad --append --write > $out/autotune_wgrad_x3_vox64_B16.txt 2>&1; rc=$?
echo "autotune x3 rc=$rc: $(grep 'conv kernels per step' $out/autotune_wgrad_x3_vox64_B16.txt)"; [ $rc -ge 124 ] && exit $rc
cp $T $out/conv_tune_table.inc
make -C optimalstrategiesagainstgenerativeattacks_amd/csrc -j8 > $out/make.log 2>&1 || { echo make failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_tuned_rows.py -m gpu -q > $out/pytest_tuned.log 2>&1; echo "tuned rows rc=$?"; tail -2 $out/pytest_tuned.log
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1; echo "shapes rc=$?"; tail -1 $out/conv_shapes_fp32.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16x3 --no-traffic --no-kernel-bench > $GRAFT_REPO_ROOT/$out/stats.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --no-cpu-baseline > $out/bench.log 2>&1; echo "bench rc=$?: $(grep -o '"value": [0-9.]*, "unit"' $out/bench.log | head -1) $(grep -o '"bf16x3_path": {"value": [0-9.]*' $out/bench.log)"
