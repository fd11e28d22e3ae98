#!/bin/bash
# the round-4 experiment tables again on round 5's final code (the part of tools/measure_round.sh that QUICK=1 skips)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r05
timeout 600 python tools/mp_ablate.py --exp variants,cap,trace > gpurun_out/${tag}_mp_ablation.txt 2>&1
timeout 600 python tools/mp_ablate.py --exp data > gpurun_out/${tag}_mp_data_dependence.txt 2>&1
timeout 600 python tools/bn_tune.py > gpurun_out/${tag}_bn_passes.txt 2>&1
timeout 600 python tools/pw_ablate.py > gpurun_out/${tag}_pw_ablation_trace.txt 2>&1
{
  timeout 600 python tools/step_ab.py --rounds 4 --forward --ab conv_pw=RYOLO_CONV1X1: --ab igemm_1x1=RYOLO_CONV1X1:igemm
  timeout 600 python tools/step_ab.py --rounds 3 --steps 1 --forward --ab fused_heads_and_stem_pair=RYOLO_HEAD_DECODE:1,RYOLO_STEM_PAIR:1 --ab one_launch_per_layer=RYOLO_HEAD_DECODE:0,RYOLO_STEM_PAIR:0 --forward-only
  timeout 600 python tools/step_ab.py --rounds 4 --ab conv0_one_pass=RYOLO_CONV0_ONE_PASS:1 --ab conv0_two_pass_plus_wgrad=RYOLO_CONV0_ONE_PASS:0
  timeout 600 python tools/step_ab.py --rounds 4 --ab bn_reduce_in_tile_dgrads=RYOLO_BN_REDUCE_TILES:1 --ab bn_reduce_separate_pass=RYOLO_BN_REDUCE_TILES:0
  timeout 600 python tools/step_ab.py --rounds 4 --ab stem_dgrads_one_launch=RYOLO_STEM_DGRAD:3 --ab parity_class_launches=RYOLO_STEM_DGRAD:0
} > gpurun_out/${tag}_ab_round4_switches.txt 2>&1
tail -n 4 gpurun_out/${tag}_mp_ablation.txt gpurun_out/${tag}_mp_data_dependence.txt gpurun_out/${tag}_bn_passes.txt gpurun_out/${tag}_pw_ablation_trace.txt
cat gpurun_out/${tag}_ab_round4_switches.txt
