#!/bin/bash
# Builds timing-experiment variants of the library (results are WRONG by construction): which part of the implicit-GEMM
# main loop costs what.  Outputs tools/micro/libgim_dbg_<flags>.so; run with GIM_LIB_PATH=... tools/kernel_probe.py
cd "$(dirname "$0")/../../optimalstrategiesagainstgenerativeattacks_amd/csrc"
for v in NOLOAD NOSTORE NOLDS "NOLOAD NOSTORE" "NOLOAD NOSTORE NOBARRIER" "NOLOAD NOSTORE NOBARRIER NOLDS"; do
  flags=""; tag=""
  for f in $v; do flags="$flags -DGIM_DBG_$f"; tag="${tag}_$f"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags conv_igemm.hip spectral.hip norm.hip pointwise.hip gemm.hip adam.hip -o ../../tools/micro/libgim_dbg$tag.so 2>/dev/null &
done
wait
ls -la ../../tools/micro/*.so
