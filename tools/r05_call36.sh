#!/bin/bash
# NOTE: the environment knob this A/B used existed only for the experiment (results: profiles/r05_ab_log.txt)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 python tools/step_ab.py --rounds 4 --ab wgrad_1x1_target_320=RYOLO_WGRAD_TARGET_1X1:320 --ab wgrad_1x1_target_256=RYOLO_WGRAD_TARGET_1X1:256 --ab wgrad_1x1_target_512=RYOLO_WGRAD_TARGET_1X1:512 > gpurun_out/r05_ab_call36.txt 2>&1
cat gpurun_out/r05_ab_call36.txt
