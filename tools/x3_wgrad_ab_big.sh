#!/bin/bash
b() { for i in 1 2; do GIM_CONV_PREC=1 python bench.py $BARGS --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-bench 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print(l['value'], end='  ')"; done; echo; }
for BARGS in "--batch 64" "--workload om32" "--workload vox128"; do
  export BARGS
  echo -n "$BARGS  x3 wgrad off: "; GIM_X3_NO_WGRAD=1 b
  echo -n "$BARGS  x3 wgrad on:  "; b
  echo -n "$BARGS  x3 wgrad on, 1 WG/CU: "; GIM_LIB_PATH=$PWD/tools/micro/libgim_dbg_OCC1.so b
done
python tools/host_overhead.py 2>&1 | tail -5
