#!/bin/bash
# Builds timing-experiment variants of the library (results are WRONG by construction): which part of the implicit-GEMM
# main loop costs what.  Outputs tools/micro/libgim_dbg_<flags>.so; run with GIM_LIB_PATH=... tools/kernel_probe.py
#   tools/micro/build_dbg.sh                      -> the fp32-kernel set (run_dbg.sh)
#   tools/micro/build_dbg.sh x3                   -> the bf16x3 set (run_x3dbg.sh): X3_NOSPLITB X3_NOSPLIT NOLOAD NOSTORE ...
#   tools/micro/build_dbg.sh A+B C ...            -> one library per argument, flags joined by '+' (-DGIM_DBG_A -DGIM_DBG_B)
cd "$(dirname "$0")/../../optimalstrategiesagainstgenerativeattacks_amd/csrc"
if [ $# -eq 0 ]; then set -- NOLOAD NOSTORE NOLDS NOLOAD+NOSTORE NOLOAD+NOSTORE+NOBARRIER NOLOAD+NOSTORE+NOBARRIER+NOLDS; fi
if [ "$1" = x3 ]; then set -- X3_NOSPLITB X3_NOSPLIT NOLOAD NOSTORE NOLOAD+NOSTORE NOLOAD+NOSTORE+NOBARRIER; fi
for v in "$@"; do
  flags=""
  for f in ${v//+/ }; do flags="$flags -DGIM_DBG_$f"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags conv_igemm.hip spectral.hip norm.hip pointwise.hip gemm.hip adam.hip -o ../../tools/micro/libgim_dbg_$v.so 2>/dev/null &
done
wait
ls -la ../../tools/micro/*.so
