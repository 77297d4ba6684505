#!/bin/bash
# HBM fetch bytes and time of ONE conv launch under the K orders of the fast path (GIM_CONV_KGROUP = 0 tap-major, G = channel-group-major)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_korder
mkdir -p $out
cd $R
for prec in 0 1; do
  for kg in 0 16 32 64; do
    for shape in "fwd 320 64 64 64 3" "dgrad 320 64 64 64 3" "fwd 160 16 128 256 3" "fwd 160 8 512 512 3"; do
      echo -n "prec=$prec kgroup=$kg: "; GIM_CONV_PREC=$prec GIM_CONV_KGROUP=$kg python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for kg in 0 16 32 64; do
  GIM_CONV_KGROUP=$kg rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/k${kg} -o r -- python3 $R/tools/kernel_probe.py fwd 320 64 64 64 3 0 3 > $out/k${kg}.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob
for kg in (0,16,32,64):
    f=glob.glob("$out/k%d/*counter_collection.csv"%kg)[0]
    vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "conv_igemm" in r["Kernel_Name"]]
    print("kgroup %2d: fetch %.1f MB per launch (x2 gfx950 correction applied); input 335.5 MB"%(kg, sum(vals)/len(vals)*1024*2/1e6))
PY
