#!/bin/bash
# HBM fetch / write bytes of ONE conv launch under the three tile orders (GIM_CONV_XCD_MODE 0/1/2)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_xcd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GIM_CONV_XCD_MODE=$mode rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/m${mode}_$c -o r -- python3 $R/tools/kernel_probe.py fwd 320 64 64 64 3 0 3 > $out/m${mode}_$c.log 2>&1 || exit 1
  done
done
python3 - <<PY
import csv, glob, collections
for mode in (0,1,2):
    res={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        f=glob.glob("$out/m%d_%s/*counter_collection.csv"%(mode,c))[0]
        vals=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "conv_igemm" in r["Kernel_Name"]]
        res[c]=sum(vals)/len(vals)
    print("xcd mode %d: fetch %.1f MB (x2 gfx950 correction applied), write %.1f MB per launch; algorithmic 335.5 + 335.5 MB"%(mode, res["FETCH_SIZE"]*1024*2/1e6, res["WRITE_SIZE"]*1024/1e6))
PY
