#!/bin/bash
# flake hunt on the final code: the whole GPU tier twice, the soak file once more
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for i in 1 2; do
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -12 > gpurun_out/r05_flake_$i.log 2>&1
done
timeout 600 python -m pytest tests/test_soak_counted_waits_gpu.py -q 2>&1 | tail -3 > gpurun_out/r05_flake_soak.log 2>&1
cat gpurun_out/r05_flake_1.log gpurun_out/r05_flake_2.log gpurun_out/r05_flake_soak.log
