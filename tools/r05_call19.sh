#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tools/step_ab.py --rounds 6 --forward --forward-only --ab nt_min_128MiB_default=RYOLO_NT_OUT_MIN_MB:128 --ab nt_min_90MiB=RYOLO_NT_OUT_MIN_MB:90 --ab nt_min_40MiB=RYOLO_NT_OUT_MIN_MB:40 > gpurun_out/r05_ab_call19.txt 2>&1
cat gpurun_out/r05_ab_call19.txt
