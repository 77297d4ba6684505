#!/bin/bash
cd "$(dirname "$0")/../.."
for shape in "fwd 320 64 64 64 3" "fwd 160 16 128 256 3" "dgrad 160 16 128 256 3"; do
  echo "== $shape"
  echo -n "base: "; python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
  for lib in tools/micro/libgim_dbg_*.so; do
    echo -n "$(basename $lib .so | sed s/libgim_dbg_//): "; GIM_LIB_PATH=$PWD/$lib python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
  done
done
