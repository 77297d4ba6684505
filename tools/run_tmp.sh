#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02v; mkdir -p $out
Q="--no-cpu-baseline --no-kernel-bench --no-traffic --no-bf16x3"
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $Q > $out/bench_$name.log 2>&1
  echo "$name: $(grep -o '"value": [0-9.]*, "unit"' $out/bench_$name.log | head -1) median $(grep -o '"ms_per_step_median": [0-9.]*' $out/bench_$name.log) steps $(grep -o '"per_step_ms": [^]]*]' $out/bench_$name.log | cut -c1-140)"
}
run default1 A=1
run rccl1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 GIM_FORCE_ALLREDUCE=1
run default2 A=1
run rccl2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29552 GIM_FORCE_ALLREDUCE=1
run rccl_q4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29553 GIM_FORCE_ALLREDUCE=1 GPU_MAX_HW_QUEUES=4
run gloo2 GIM_BENCH_BACKEND=gloo GIM_BENCH_ONE_DEVICE=1 X=1
