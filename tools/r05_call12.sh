#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_wide_tile_variants_gpu.py tests/test_model_gpu.py -q -s -k "mp or mq or wide or configs1_bs32" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r05_call12_tests.log 2>&1
{
  timeout 600 python tools/step_ab.py --rounds 5 --forward --ab k_tap_major=RYOLO_MQ_KORDER:0 --ab k_channel_major_from_512=RYOLO_X_UNUSED:1 2>&1
} > gpurun_out/r05_ab_call12.txt 2>&1
cat gpurun_out/r05_call12_tests.log gpurun_out/r05_ab_call12.txt
