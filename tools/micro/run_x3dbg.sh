#!/bin/bash
cd $GRAFT_REPO_ROOT
export GIM_CONV_PREC=1
for shape in "fwd 320 64 64 64 3" "fwd 160 16 128 256 3" "fwd 160 8 512 512 3"; do
  echo "== $shape"
  echo -n "base x3: "; python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
  echo -n "base f32: "; GIM_CONV_PREC=0 python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
  for t in 128 641 1264 64; do echo -n "x3 tile $t: "; GIM_CONV_TILE=$t python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1; done
  for lib in tools/micro/libgim_dbg_*.so; do
    echo -n "$(basename $lib .so | sed s/libgim_dbg_//): "; GIM_LIB_PATH=$PWD/$lib python tools/kernel_probe.py $shape 0 10 2>&1 | tail -1
  done
done
