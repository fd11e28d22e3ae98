#!/bin/bash
# NOTE: the environment knob(s) this A/B used existed only for the experiment and were removed with it (results: profiles/r05_ab_log.txt);
# re-running this script on the current tree compares identical settings.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py --rounds 5 --ab base=RYOLO_BN_RED_CT:32,RYOLO_WGRAD_TARGET:0 --ab bn_red_ct16=RYOLO_BN_RED_CT:16,RYOLO_WGRAD_TARGET:0 --ab bn_red_ct8=RYOLO_BN_RED_CT:8,RYOLO_WGRAD_TARGET:0 --ab wgrad_target_384=RYOLO_BN_RED_CT:32,RYOLO_WGRAD_TARGET:384 --ab wgrad_target_256=RYOLO_BN_RED_CT:32,RYOLO_WGRAD_TARGET:256 --ab wgrad_target_768=RYOLO_BN_RED_CT:32,RYOLO_WGRAD_TARGET:768 > gpurun_out/r05_ab_call32.txt 2>&1
cat gpurun_out/r05_ab_call32.txt
