#!/bin/bash
# What engine clock and board power does the default bench run at?  Samples rocm-smi beside bench.py (one GPU process + one reader).
#   tools/clock_under_load.sh <out-dir>     (on the GPU box; the MFMA peak of MI355X_MICROARCH.md is quoted at 2.4 GHz)
out=${1:-gpurun_out/clock}
mkdir -p $out
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
rocm-smi --showclocks --showpower --showtemp > $out/idle.txt 2>&1
python bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-kernel-bench --no-traffic > $out/bench.log 2> $out/bench.err &
pid=$!
sleep 7   # import + warm-up
: > $out/samples.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|Sensor junction" >> $out/samples.txt
  echo "--" >> $out/samples.txt
  sleep 0.5
done
wait $pid
grep -o '"value": [0-9.]*, "unit"' $out/bench.log | head -1
python3 - $out/samples.txt <<'PY'
import re, sys
s = open(sys.argv[1]).read()
clk = [int(x) for x in re.findall(r"sclk clock level: \w+: \((\d+)Mhz\)", s)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", s)]
if clk:
    print("sclk samples %d: mean %.0f MHz, min %d, max %d" % (len(clk), sum(clk) / len(clk), min(clk), max(clk)))
tj = [float(x) for x in re.findall(r"Sensor junction\) \(C\): ([\d.]+)", s)]
if tj:
    print("junction temperature: mean %.0f C, max %.0f C" % (sum(tj) / len(tj), max(tj)))
if pw:
    print("power samples %d: mean %.0f W, max %.0f W" % (len(pw), sum(pw) / len(pw), max(pw)))
PY
