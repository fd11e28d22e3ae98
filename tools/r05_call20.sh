#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_engine_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_train_fullsize_gpu.py -q -k "wgrad or reduce or reproducible or second_step or reducer or two_ranks or batched or composed" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -15 > gpurun_out/r05_call20_tests.log 2>&1
timeout 600 python tools/step_ab.py --rounds 6 --ab reduce_per_layer=RYOLO_WGRAD_BATCH_REDUCE:0 --ab reduce_one_launch_per_segment=RYOLO_WGRAD_BATCH_REDUCE:1 > gpurun_out/r05_ab_call20.txt 2>&1
cat gpurun_out/r05_call20_tests.log gpurun_out/r05_ab_call20.txt
