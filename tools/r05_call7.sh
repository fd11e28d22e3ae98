#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_wide_tile_variants_gpu.py tests/test_soak_counted_waits_gpu.py -q -k "mp or mq or wide" 2>&1 | tail -5 > gpurun_out/r05_call7_tests.log 2>&1
for sw in 1 2; do
  RYOLO_MQ_SWEEP=$sw bash tools/traffic_pmc.sh traffic_sweep$sw 3 1 128 256 76 5 9 > gpurun_out/traffic_sweep$sw.log 2>&1
  find gpurun_out/traffic_sweep$sw -name "*.db" -delete
done
{
  timeout 600 python tools/step_ab.py --rounds 5 --forward --ab sweep_one_per_half=RYOLO_MQ_SWEEP:1 --ab sweep_two_blocks=RYOLO_MQ_SWEEP:2 2>&1
} > gpurun_out/r05_ab_call7.txt 2>&1
cat gpurun_out/r05_call7_tests.log; tail -n 9 gpurun_out/traffic_sweep1.log gpurun_out/traffic_sweep2.log; cat gpurun_out/r05_ab_call7.txt
