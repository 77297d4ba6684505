#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02i; mkdir -p $out
P="timeout -k 10 120 python tools/kernel_probe.py"
{
for kind in fwd dgrad; do
 for shp in "80 4 512 512 3 0 20 0" "80 8 256 256 3 0 20 0" "80 16 128 128 3 0 20 0" "16 32 64 128 3 0 20 0" "16 16 128 256 3 0 20 0" "80 32 64 64 3 0 20 0" "80 8 512 512 3 0 20 1" "16 8 512 512 3 0 20 1"; do
  for ks in 1 2 4 8; do
   for tile in 64 6432; do
     $P $kind $shp 0 $tile $ks 2>&1 | grep -v amdgpu.ids | sed "s/^/tile=$tile ks=$ks: /" || exit 1
   done
  done
 done
done
} > $out/kb32_probe.txt 2>&1
tail -4 $out/kb32_probe.txt
{
for cin in 32 64 128 256 512; do $P fwd 80 32 $cin 128 3 0 20 0 0 641 1 2>&1 | grep -v amdgpu.ids; done
for cin in 32 64 128 256 512; do $P fwd 320 32 $cin 128 3 0 10 0 0 641 1 2>&1 | grep -v amdgpu.ids; done
} > $out/time_vs_k.txt 2>&1
cat $out/time_vs_k.txt
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $out/pytest_ops.log 2>&1; echo "ops tests: $(tail -1 $out/pytest_ops.log)"
