#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_train_ops_gpu.py -q -k "mp or mq or stats" 2>&1 | tail -4 > gpurun_out/r05_call8_tests.log 2>&1
{
  # write-through (sc1) stores for conv outputs below the non-temporal threshold, in the chain (bs-32 forward)
  timeout 600 python tools/step_ab.py --rounds 5 --forward --forward-only --ab default_write_back=RYOLO_SC1_OUT_MIN_MB:1000000 --ab sc1_from_32MiB=RYOLO_SC1_OUT_MIN_MB:32 --ab sc1_from_8MiB=RYOLO_SC1_OUT_MIN_MB:8 --ab sc1_everything=RYOLO_SC1_OUT_MIN_MB:0 2>&1
  # the two-block store order in the statistics epilogue (training forward) -- step A/B
  timeout 600 python tools/step_ab.py --rounds 5 --ab stats_sweep_one_per_half=RYOLO_MQ_SWEEP_STATS:1 --ab stats_sweep_two_blocks=RYOLO_MQ_SWEEP_STATS:2 2>&1
} > gpurun_out/r05_ab_call8.txt 2>&1
cat gpurun_out/r05_call8_tests.log gpurun_out/r05_ab_call8.txt
