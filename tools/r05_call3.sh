#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 900 python -m pytest tests/test_train_engine_gpu.py tests/test_model_gpu.py -q -s -k "configs3_bs64 or configs1_bs32" 2>&1 | grep -v "^$" | tail -40
} > gpurun_out/r05_call3_tests.log 2>&1
timeout 300 python tools/mq128_bench.py --no-train > gpurun_out/r05_mq128_wgpc.txt 2>&1
timeout 300 python tools/step_small_ops.py > gpurun_out/r05_step_small_ops.txt 2>&1
cat gpurun_out/r05_call3_tests.log gpurun_out/r05_mq128_wgpc.txt gpurun_out/r05_step_small_ops.txt
