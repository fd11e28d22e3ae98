#!/bin/bash
# round 5: the batched split-K reduce's four-channel form (job kind 3): bitwise tests, then the in-process A/B
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_engine_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_train_fullsize_gpu.py -q -k "wgrad or reduce or reproducible or second_step or reducer or two_ranks or batched or composed or no_graph" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -15 > gpurun_out/r05_call22_tests.log 2>&1
timeout 600 python tools/step_ab.py --rounds 6 --ab reduce_one_group_per_workgroup=RYOLO_WGRAD_REDUCE_V4:2 --ab reduce_wide_workgroups=RYOLO_WGRAD_REDUCE_V4:3 > gpurun_out/r05_ab_call22.txt 2>&1
cat gpurun_out/r05_call22_tests.log gpurun_out/r05_ab_call22.txt
