#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02g; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -k "conv or linear or wgrad or trainer_protocol or loop_golden" > $out/pytest_sub.log 2>&1; rc=$?
echo "pytest sub rc=$rc: $(grep -E 'passed|failed' $out/pytest_sub.log | tail -1)"; grep -E "^FAILED|^ERROR" $out/pytest_sub.log | head
[ $rc -ge 124 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_tuned_rows.py -m gpu -q -x > $out/pytest_tuned.log 2>&1; rc=$?; echo "tuned rows rc=$rc: $(tail -1 $out/pytest_tuned.log)"
[ $rc -ge 124 ] && exit $rc
Q="--no-cpu-baseline --no-kernel-bench --no-traffic"
timeout -k 10 300 python tools/conv_shapes_bench.py > $out/conv_shapes_fp32.txt 2>&1; tail -1 $out/conv_shapes_fp32.txt
for i in 1 2; do timeout -k 10 300 python bench.py $Q > $out/bench_$i.log 2>&1; echo "bench $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$i.log | head -1) $(grep -o '"bf16x3_path": {"value": [0-9.]*' $out/bench_$i.log)"; done
for i in 1 2; do MASTER_ADDR=127.0.0.1 MASTER_PORT=2955$i GIM_FORCE_ALLREDUCE=1 timeout -k 10 300 python bench.py $Q --no-bf16x3 > $out/bench_rccl_$i.log 2>&1; echo "rccl1 $i: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_rccl_$i.log | head -1) $(grep -o '"per_step_ms": [^]]*]' $out/bench_rccl_$i.log)"; done
GIM_BENCH_BACKEND=gloo GIM_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 2 $Q --no-bf16x3 --steps 10 > $out/bench_gloo2.log 2>&1; echo "gloo 2 ranks on one card rc=$?: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_gloo2.log | head -1) $(grep -o '"rccl_ranks": [0-9]*' $out/bench_gloo2.log)"
