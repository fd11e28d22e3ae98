#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -q -k "channel_major" 2>&1 | tail -4 > gpurun_out/r05_call13_tests.log 2>&1
{
  timeout 900 python tools/step_ab.py --rounds 6 --forward --ab k_tap_major=RYOLO_MQ_KORDER_MIN_CIN:100000 --ab k_channel_major_from_512=RYOLO_MQ_KORDER_MIN_CIN:512 --ab k_channel_major_from_256=RYOLO_MQ_KORDER_MIN_CIN:256 --ab k_channel_major_from_128=RYOLO_MQ_KORDER_MIN_CIN:128 2>&1
} > gpurun_out/r05_ab_call13.txt 2>&1
cat gpurun_out/r05_call13_tests.log gpurun_out/r05_ab_call13.txt
