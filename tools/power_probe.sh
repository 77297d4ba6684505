#!/bin/bash
# Clock / power samples of the GPU while the step runs on the fp32-MFMA path and on the bf16x3 path (is the step-time spread of
# the bf16x3 path a power / clock effect?)
cd $GRAFT_REPO_ROOT
for prec in 0 1; do
  GIM_CONV_PREC=$prec python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-kernel-bench --no-bf16x3 > /tmp/b_$prec.log 2>&1 &
  pid=$!
  sleep 6
  for i in $(seq 12); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Power (W)\|Average Graphics Package Power\|Socket Power" | tr '\n' ' ' | sed 's/=\+//g' | cut -c1-220; echo
    sleep 0.4
  done
  wait $pid
  echo "prec=$prec: $(grep -o '"value": [0-9.]*' /tmp/b_$prec.log | head -1)"
done
