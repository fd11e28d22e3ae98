#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_engine_gpu.py -q -s -k "configs3_bs64" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r05_call4_tests.log 2>&1
timeout 300 python tools/step_small_ops.py > gpurun_out/r05_step_small_ops.txt 2>&1
cat gpurun_out/r05_call4_tests.log gpurun_out/r05_step_small_ops.txt
