#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_contract_gpu.py -q -x  2>&1 | tail -30 > gpurun_out/r05_call14_tests.log 2>&1
cat gpurun_out/r05_call14_tests.log
