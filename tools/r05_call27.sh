#!/bin/bash
# NOTE: the environment knob(s) this A/B used existed only for the experiment and were removed with it (results: profiles/r05_ab_log.txt);
# re-running this script on the current tree compares identical settings.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "loss or fused or train_step or reproducible or no_graph" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -15 > gpurun_out/r05_call27_tests.log 2>&1
timeout 800 python tools/step_ab.py --rounds 6 --ab bitmap_per_chunk_vector_loads=RYOLO_LOSS_BITMAP_WINDOW:0 --ab bitmap_window_scalar_loads=RYOLO_LOSS_BITMAP_WINDOW:1 > gpurun_out/r05_ab_call27.txt 2>&1
cat gpurun_out/r05_call27_tests.log gpurun_out/r05_ab_call27.txt
