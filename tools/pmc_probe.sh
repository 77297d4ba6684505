#!/bin/bash
# PMC passes (one counter set per run, kernel-trace only) on single conv shapes.  usage: tools/pmc_probe.sh <outdir>
# the HIP runtime reads this when it starts - under rocprofv3 --pmc the profiler initialises the GPU before python imports the package, so set it here
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in "fwd 320 64 64 64 3" "dgrad 320 64 64 64 3" "wgrad 320 64 64 64 3" "fwd 80 32 64 128 3" "fwd 160 16 128 256 3"; do
  tag=$(echo $shape | tr ' ' '_')
  for pmc in "MfmaUtil" "MeanOccupancyPerActiveCU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
    ptag=$(echo $pmc | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/${tag}__${ptag} -o r -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py $shape 0 3 > $out/${tag}__${ptag}.log 2>&1 || echo "FAILED $tag $pmc"
  done
done
echo done
