#!/bin/bash
# round 5, second GPU call: the whole GPU tier (128-channel tiles opt-in now, bs-64 step test against the ATen yardstick), the in-chain
# attribution passes of the wide 3x3 kernel, and the in-chain sweep of the non-temporal output threshold
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== whole GPU tier"; date
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 2>&1 | tail -60
} > gpurun_out/r05_call2_gputests.log 2>&1
bash tools/inchain.sh 14 > gpurun_out/r05_inchain_L14.txt 2>&1
cp gpurun_out/inchain/inchain_summary.json gpurun_out/r05_inchain_L14.json 2>/dev/null
{
  date
  timeout 600 python tools/step_ab.py --rounds 4 --forward --forward-only --ab nt_min_32MiB=RYOLO_NT_OUT_MIN_MB:32 --ab nt_min_64MiB=RYOLO_NT_OUT_MIN_MB:64 --ab nt_min_128MiB_default=RYOLO_NT_OUT_MIN_MB:128 --ab nt_min_256MiB=RYOLO_NT_OUT_MIN_MB:256 --ab nt_never=RYOLO_NT_OUT_MIN_MB:1000000 2>&1
} > gpurun_out/r05_ab_call2.txt 2>&1
tail -n 8 gpurun_out/r05_call2_gputests.log
cat gpurun_out/r05_inchain_L14.txt gpurun_out/r05_ab_call2.txt
