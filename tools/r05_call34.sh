#!/bin/bash
# round 5: the GPU tier once more on the final code (flake hunt), the way the driver runs it (-x)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_flake_full.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r05_flake_full.log | tail -8
