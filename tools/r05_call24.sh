#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 800 python tools/step_ab.py --rounds 5 --ab slab_min_256=RYOLO_BN_SLAB_MIN:256,RYOLO_BN_SLABS:1024 --ab slab_min_128=RYOLO_BN_SLAB_MIN:128,RYOLO_BN_SLABS:1024 --ab slab_min_96=RYOLO_BN_SLAB_MIN:96,RYOLO_BN_SLABS:1024 --ab slab_min_160=RYOLO_BN_SLAB_MIN:160,RYOLO_BN_SLABS:1024 --ab slab_min_128_slabs_2048=RYOLO_BN_SLAB_MIN:128,RYOLO_BN_SLABS:2048 --ab slab_min_128_slabs_512=RYOLO_BN_SLAB_MIN:128,RYOLO_BN_SLABS:512 > gpurun_out/r05_ab_call24b.txt 2>&1
cat gpurun_out/r05_ab_call24b.txt
