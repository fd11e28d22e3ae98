#!/bin/bash
# two counter passes over the bs-64 train step (5 traced steps: 3 warm-up + 2 timed), per-kernel summary -> gpurun_out/r05_step_traffic.txt
set -u
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/step_traffic
rm -rf $out; mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o t -- python $root/bench.py --mode train --bs 64 --steps 2 --warmup 3 --no-cpu-baseline --no-nms --no-kernel-table > $out/$c.log 2>&1 || echo "$c pass failed"
done
cd $root
python tools/step_traffic.py $out 5 > gpurun_out/r05_step_traffic.txt 2>&1
find $out -name "*.csv" -size +1M -delete
cat gpurun_out/r05_step_traffic.txt
