#!/bin/bash
# round 5, first GPU call: the new kernels' tests first (short timeout: a hung kernel must not eat the budget), then the whole GPU tier,
# then the per-layer table of the 128-channel tiles and the step / forward A/Bs.  Everything lands under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== new-kernel tests"; date
  timeout 600 python -m pytest tests/test_conv_gpu.py -q -k "mq128" --maxfail=8 2>&1 | tail -25
  timeout 600 python -m pytest tests/test_train_ops_gpu.py -q -k "dgrad_vs_autograd or folded_bn_reduce or 128_channel_mq" --maxfail=8 2>&1 | tail -25
} > gpurun_out/r05_call1_new.log 2>&1
{
  echo "== whole GPU tier"; date
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 2>&1 | tail -60
} > gpurun_out/r05_call1_gputests.log 2>&1
{
  date
  timeout 600 python tools/mq128_bench.py 2>&1
} > gpurun_out/r05_mq128_bench.txt 2>&1
{
  date
  timeout 600 python tools/step_ab.py --rounds 4 --forward --ab mq128_off=RYOLO_MQ128:0 --ab mq128_3x3=RYOLO_MQ128:1 --ab mq128_3x3_and_1x1=RYOLO_MQ128:2 2>&1
  timeout 600 python tools/step_ab.py --rounds 4 --ab wgrad_main_stream=RYOLO_WGRAD_STREAM:0 --ab wgrad_side_stream=RYOLO_WGRAD_STREAM:1 2>&1
} > gpurun_out/r05_ab_call1.txt 2>&1
tail -5 gpurun_out/r05_call1_new.log gpurun_out/r05_call1_gputests.log
cat gpurun_out/r05_mq128_bench.txt gpurun_out/r05_ab_call1.txt
