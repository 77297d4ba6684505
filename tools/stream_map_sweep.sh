#!/bin/bash
# Which stream roles may run concurrently?  Runs bench.py (vox64, 16 episodes) for a list of role->stream maps
# (GIM_STREAM_MAP, see gim_img_models.py), with N hardware queues (GPU_MAX_HW_QUEUES), with / without a one-rank RCCL
# communicator (GIM_FORCE_ALLREDUCE).  Add --no-traffic when sweeping.
#   usage: tools/stream_map_sweep.sh OUT "Q1 Q2" "map1 map2 ..." ["0 1"]
OUT=${1:-gpurun_out/stream_map.txt}
QS=${2:-"4 8"}
MAPS=${3:-"0,1,2,2,0"}
RCCLS=${4:-"0 1"}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
: > "$OUT"
for q in $QS; do
  for m in $MAPS; do
    for r in $RCCLS; do
      if [ "$r" = 1 ]; then export GIM_FORCE_ALLREDUCE=1; else unset GIM_FORCE_ALLREDUCE; fi
      v=$(GPU_MAX_HW_QUEUES=$q GIM_STREAM_MAP=$m timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-bench --no-traffic 2>/dev/null \
          | grep -o '"value": [0-9.]*' | head -1 | tr '\n' ' ')
      echo "Q=$q map=$m rccl=$r  $v" | tee -a "$OUT"
    done
  done
done
