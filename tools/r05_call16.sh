#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 4 --share-gpu --dist-backend gloo --steps 2 --warmup 1 --bs 2 --size 160 --train-bs 2 --no-nms > gpurun_out/r05_four_ranks_one_gpu_line.json 2> gpurun_out/r05_four_ranks.err
echo "rc=$?"; head -c 3000 gpurun_out/r05_four_ranks_one_gpu_line.json; echo; tail -n 5 gpurun_out/r05_four_ranks.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --no-graph --steps 2 --warmup 1 --bs 2 --size 160 --train-bs 2 --no-nms > gpurun_out/r05_two_ranks_nograph_line.json 2> gpurun_out/r05_two_ranks_nograph.err
echo "rc=$?"; python3 -c "
import json
d=json.load(open('gpurun_out/r05_two_ranks_nograph_line.json')); print(d['launch_mode'], d['value'], d['allreduce'])"
