#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02af; mkdir -p $out
for wl in "vox64 16" "om32 32" "vox64 64" "vox128 2"; do
  set -- $wl
  timeout -k 10 900 python tools/conv_autotune.py --workload $1 --batch $2 --kinds fwd,dgrad --write --out $out/rows_$1_B$2.inc > $out/autotune_$1_B$2.txt 2>&1 || { echo "autotune $wl failed"; tail -5 $out/autotune_$1_B$2.txt; exit 1; }
  echo "$wl: $(grep 'conv kernels per step' $out/autotune_$1_B$2.txt) rows $(grep -c '^    {' $out/rows_$1_B$2.inc)"
done
