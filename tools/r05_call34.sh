#!/bin/bash
# round 5: the GPU tier twice more on the final code (flake hunt), the way the driver runs it (-x)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for k in 1 2; do
  timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05_flake_full_$k.log
done
tail -n 3 gpurun_out/r05_flake_full_1.log gpurun_out/r05_flake_full_2.log
