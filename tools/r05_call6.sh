#!/bin/bash
# round 5: the whole GPU tier on the final code, then the measurement pipeline (QUICK: driver-facing lines, rocprof summaries, traffic, r05 tables)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== whole GPU tier"; date
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 2>&1 | tail -40
} > gpurun_out/gputests_r05_final.log 2>&1
QUICK=1 bash tools/measure_round.sh r05 > gpurun_out/measure_r05.log 2>&1
tail -n 6 gpurun_out/gputests_r05_final.log
head -c 1500 gpurun_out/r05_bench_line.json; echo
tail -n 5 gpurun_out/measure_r05.log
