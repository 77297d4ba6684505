#!/bin/bash
# Whole-step effect of the conv K order (GIM_CONV_KGROUP): episodes/s (fp32 MFMA, bf16x3) and FETCH_SIZE of the step
R=$GRAFT_REPO_ROOT
cd $R
for kg in 0 32; do
  for i in 1 2; do
    echo -n "kgroup=$kg: "; GIM_CONV_KGROUP=$kg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-bench 2>/dev/null | grep -o '"value": [0-9.]*' | head -2 | tr '\n' ' '; echo
  done
done
cd /tmp && export TMPDIR=/tmp
for kg in 0 32; do
  GIM_CONV_KGROUP=$kg rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/korder_step/k$kg -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-bench --no-bf16x3 > $R/gpurun_out/korder_step_k$kg.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob
for kg in (0,32):
    f=glob.glob("$R/gpurun_out/korder_step/k%d/*counter_collection.csv"%kg)[0]
    tot=sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)))
    print("kgroup %2d: fetch %.1f GB per step (3 steps in the run, x2 correction)"%(kg, tot*1024*2/3/1e9))
PY
