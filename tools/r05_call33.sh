#!/bin/bash
# NOTE: the environment knob(s) this A/B used existed only for the experiment and were removed with it (results: profiles/r05_ab_log.txt);
# re-running this script on the current tree compares identical settings.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py --rounds 6 --ab bn_reduce_4_pixels_per_trip=RYOLO_BN_RED_PT:4 --ab bn_reduce_8_pixels_per_trip=RYOLO_BN_RED_PT:8 > gpurun_out/r05_ab_call33.txt 2>&1
cat gpurun_out/r05_ab_call33.txt
