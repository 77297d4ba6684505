#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02t; mkdir -p $out
P="timeout -k 10 120 python tools/kernel_probe.py"
{
for lib in "" tools/micro/_exp/libgim_stagger1.so tools/micro/_exp/libgim_stagger3.so; do
 for shp in "fwd 80 32 64 128 3 0 20 0" "dgrad 80 32 64 128 3 0 20 0" "fwd 160 64 64 64 3 0 20 1" "fwd 80 16 128 256 3 0 20 0" "fwd 80 64 64 64 9 0 10 1" "fwd 320 32 128 128 3 0 10 0"; do
   GIM_LIB_PATH=$lib $P $shp 2>&1 | grep -v amdgpu.ids | sed "s#^#[${lib:-product}] #" || exit 1
 done
done
} > $out/stagger_probe.txt 2>&1
cat $out/stagger_probe.txt | cut -c1-60,150-230
