#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/mp_ablate.py --exp mq > gpurun_out/r05_mp_vs_mq.txt 2>&1
timeout 600 python tools/pw_bench.py --train > gpurun_out/r05_pw_vs_igemm.txt 2>&1
tail -n 40 gpurun_out/r05_mp_vs_mq.txt; tail -n 12 gpurun_out/r05_pw_vs_igemm.txt
