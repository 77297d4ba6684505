#!/bin/bash
# A/B of the bf16x3 weight-gradient kernel inside the whole step: N runs of 20 steps per variant (the step time is bimodal)
N=${1:-4}
b() { for i in $(seq $N); do GIM_CONV_PREC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-bench 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print(l['value'], end='  ')"; done; echo; }
echo -n "x3 wgrad 128x128 (2 WG/CU): "; b
for lib in tools/micro/libgim_dbg_*.so; do echo -n "$(basename $lib .so | sed s/libgim_dbg_//): "; GIM_LIB_PATH=$PWD/$lib b; done
