#!/bin/bash
# NOTE: the environment knob(s) this A/B used existed only for the experiment and were removed with it (results: profiles/r05_ab_log.txt);
# re-running this script on the current tree compares identical settings.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 800 python tools/step_ab.py --rounds 5 --ab dense_blocks_8192=RYOLO_LOSS_DENSE_BLOCKS:8192 --ab dense_blocks_4096=RYOLO_LOSS_DENSE_BLOCKS:4096 --ab dense_blocks_2048=RYOLO_LOSS_DENSE_BLOCKS:2048 --ab dense_blocks_1024=RYOLO_LOSS_DENSE_BLOCKS:1024 > gpurun_out/r05_ab_call26.txt 2>&1
cat gpurun_out/r05_ab_call26.txt
