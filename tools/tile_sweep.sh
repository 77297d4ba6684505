#!/bin/bash
# per-layer tile-shape sweep (GIM_CONV_TILE: 128 = 128x128, 641 = 64x128, 1264 = 128x64, 64 = 64x64), split-K off
cd "$(dirname "$0")/.."
for shape in "80 32 64 128 3" "80 16 128 256 3" "80 8 256 512 3" "160 32 64 128 3" "160 16 128 256 3" "160 8 256 512 3" "80 64 64 64 3" "320 64 64 64 3" "80 16 256 256 3" "80 32 128 128 3"; do
  for kind in fwd dgrad; do
    for tile in 128 641 1264 64; do
      for ks in 1 2; do
        echo -n "tile=$tile ks=$ks "
        GIM_CONV_TILE=$tile GIM_CONV_KSPLIT=$ks python tools/kernel_probe.py $kind $shape 0 10 || exit 1
      done
    done
  done
done
