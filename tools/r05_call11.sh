#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -q -k "mp_tile or mq" 2>&1 | tail -6 > gpurun_out/r05_call11_tests.log 2>&1
for ko in 0 1; do
  RYOLO_MQ_KORDER=$ko bash tools/traffic_pmc.sh traffic_ko$ko 3 1 512 256 38 5 9 > gpurun_out/traffic_ko$ko.log 2>&1
  find gpurun_out/traffic_ko$ko -name "*.db" -delete
done
{
  timeout 600 python tools/step_ab.py --rounds 5 --forward --ab k_tap_major=RYOLO_MQ_KORDER:0 --ab k_channel_major=RYOLO_MQ_KORDER:1 2>&1
} > gpurun_out/r05_ab_call11.txt 2>&1
cat gpurun_out/r05_call11_tests.log; tail -n 9 gpurun_out/traffic_ko0.log gpurun_out/traffic_ko1.log; cat gpurun_out/r05_ab_call11.txt
