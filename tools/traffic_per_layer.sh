#!/bin/bash
# L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant conv shapes, one launch form per run:
#   tools/traffic_per_layer.sh <outdir under gpurun_out>;  summary: python tools/traffic_per_layer.py gpurun_out/<outdir>
#   TRAFFIC_SHAPES="80 64 6 64 9 0 4 0" tools/traffic_per_layer.sh <outdir>   - that one shape only
# the HIP runtime reads this when it starts - under rocprofv3 --pmc the profiler initialises the GPU before python imports the package, so set it here
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# N H Cin Cout K ups reps pool
SHAPES=("80 8 256 512 3 0 4 0" "80 16 128 256 3 0 4 0" "80 32 64 128 3 0 4 0" "80 8 512 512 3 0 4 1" "80 32 128 128 3 0 4 1" "80 16 256 256 3 0 4 1"
        "80 64 64 64 9 0 4 1" "160 32 64 128 3 0 4 0" "160 64 64 64 3 0 4 1" "80 64 64 64 3 0 4 1" "80 4 512 512 3 0 4 0" "80 64 6 64 9 0 4 0")
[ -n "$TRAFFIC_SHAPES" ] && SHAPES=("$TRAFFIC_SHAPES")
for shape in "${SHAPES[@]}"; do
  for kind in fwd dgrad wgrad; do
    tag=${kind}_$(echo $shape | tr ' ' '_')
    for pmc in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/$tag/$pmc -o r -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py $kind $shape > $out/${tag}_$pmc.log 2>&1 || echo "FAILED $tag $pmc"
    done
  done
  echo "$shape done"
done
