#!/bin/bash
# The measurement pipeline behind profiles/rNN_*: run on the GPU box from the repo root, e.g.
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh r02'
# Writes everything under gpurun_out/ (scratch); copy what is to be judged into profiles/.
#   <tag>_bench_line.json, <tag>_bench_per_op_events.txt   the un-profiled default bench (the line the driver also produces)
#   <tag>_bench_kernel_stats.txt    rocprofv3 --kernel-trace --stats of the default command (no CPU baseline leg)
#   <tag>_forward_kernel_stats.txt  ... of the forward leg alone (exactly the roofline kernel's launches)
#   <tag>_train_kernel_stats.txt    ... of the train step alone
#   <tag>_nms_kernel_stats.txt      ... of the 50 000-box rotated NMS
#   <tag>_train_kernel_stats.json   the same as JSON per step (bench.py embeds the newest committed one as "train_step_kernels")
#   <tag>_mp_vs_mq.txt, <tag>_mp_ablation.txt, <tag>_mp_data_dependence.txt, <tag>_bn_passes.txt   tools/mp_ablate.py / tools/bn_tune.py on the ablation build
#   <tag>_pw_vs_igemm.txt, <tag>_pw_ablation_trace.txt, <tag>_ab_log.txt   round 4: conv_pw.hip vs the 128x128 tile, its ablations / cycle stamps, step and forward A/Bs
#   traffic/traffic.json            FETCH_SIZE / WRITE_SIZE passes on the dominant layer (-> profiles/<tag>_traffic.json)
# Counter passes are separate from the kernel-trace runs (gpurun refuses --pmc combined with API traces).
set -u
tag=${1:-rXX}
root=$(pwd)
export TMPDIR=/tmp
mkdir -p $root/gpurun_out

python bench.py --dump-ops gpurun_out/${tag}_bench_per_op_events.txt > gpurun_out/${tag}_bench_line.json 2> gpurun_out/bench.err

prof() {   # prof <name> <command...>: kernel trace + stats, summarised from the rocpd database
    name=$1; shift
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/p_$name -o x -- "$@" > $root/gpurun_out/${tag}_${name}_line_profiled.json 2> $root/gpurun_out/$name.err
    cd $root
    db=$(find gpurun_out/p_$name -name "*_results.db" | head -1)
    python tools/rocpd_summary.py $db > gpurun_out/${tag}_${name}_kernel_stats.txt
    # the train-step kernel table bench.py embeds (16 traced steps = 3 warm-up + 10 timed + the 3 eager steps of the in-run kernel table)
    [ "$name" = train ] && python tools/rocpd_summary.py $db --json 16 gpurun_out/${tag}_train_kernel_stats.json
    rm -rf gpurun_out/p_$name
}
prof bench python $root/bench.py --no-cpu-baseline
prof forward python $root/bench.py --no-cpu-baseline --no-train --no-nms
prof train python $root/bench.py --mode train --bs 64 --steps 10 --warmup 3 --no-cpu-baseline --no-nms --dump-train-calls $root/gpurun_out/${tag}_train_calls.txt
prof nms python $root/tools/nms_time.py 50000 20
if [ -z "${QUICK:-}" ]; then    # QUICK=1 (round 5): the kernels these tables describe did not change; only the driver-facing lines and traffic are re-measured
# conv_mp vs conv_mq per layer, ablations (ablation build of the library: git-ignored, built here when the tree does not carry it)
[ -f rotate-yolov3_amd/libryolo_hip_ablation.so ] || python __graft_entry__.py --ablation > gpurun_out/build_ablation.log 2>&1
python tools/mp_ablate.py --exp mq > gpurun_out/${tag}_mp_vs_mq.txt 2>&1
python tools/mp_ablate.py --exp variants,cap,trace > gpurun_out/${tag}_mp_ablation.txt 2>&1
python tools/mp_ablate.py --exp data > gpurun_out/${tag}_mp_data_dependence.txt 2>&1
python tools/bn_tune.py > gpurun_out/${tag}_bn_passes.txt 2>&1

# round 4: the weight-stationary 1x1 kernel against the 128x128 tile (isolated launches, graph-replayed), where its time goes
# (ablations + cycle stamps), and the in-step / in-forward A/Bs of this round's switches
python tools/pw_bench.py --train > gpurun_out/${tag}_pw_vs_igemm.txt 2>&1
python tools/pw_ablate.py > gpurun_out/${tag}_pw_ablation_trace.txt 2>&1
{
  python tools/step_ab.py --rounds 4 --forward --ab conv_pw=RYOLO_CONV1X1: --ab igemm_1x1=RYOLO_CONV1X1:igemm
  python tools/step_ab.py --rounds 3 --steps 1 --forward --ab fused_heads_and_stem_pair=RYOLO_HEAD_DECODE:1,RYOLO_STEM_PAIR:1 --ab one_launch_per_layer=RYOLO_HEAD_DECODE:0,RYOLO_STEM_PAIR:0 --forward-only
  # the train step with / without: layer 0's one-pass backward (csrc/conv0_bwd.hip), the BatchNorm reduce folded into the one-tile data gradients
  python tools/step_ab.py --rounds 4 --ab conv0_one_pass=RYOLO_CONV0_ONE_PASS:1 --ab conv0_two_pass_plus_wgrad=RYOLO_CONV0_ONE_PASS:0
  python tools/step_ab.py --rounds 4 --ab bn_reduce_in_tile_dgrads=RYOLO_BN_REDUCE_TILES:1 --ab bn_reduce_separate_pass=RYOLO_BN_REDUCE_TILES:0
  python tools/step_ab.py --rounds 4 --ab stem_dgrads_one_launch=RYOLO_STEM_DGRAD:3 --ab parity_class_launches=RYOLO_STEM_DGRAD:0
  python tools/step_ab.py --rounds 4 --forward --ab layer0_forward_staged_in_lds=RYOLO_CONV0:halo --ab layer0_forward_direct=RYOLO_CONV0:direct
} > gpurun_out/${tag}_ab_log.txt 2>&1
fi

bash tools/traffic_pmc.sh traffic 3 1 128 256 76 2 0 > gpurun_out/traffic.log 2>&1
# the train step's dominant kernel (wgrad_wide<256,128>) on its most frequent shape, bs 64
bash tools/traffic_pmc.sh traffic_wgrad wgrad 3 1 128 256 76 4 > gpurun_out/traffic_wgrad.log 2>&1
# round 5: the 128-channel members of conv_mq's family against the tiles they would replace; the launches outside the library in one train step
python tools/mq128_bench.py > gpurun_out/${tag}_mq128_bench.txt 2>&1
python tools/step_small_ops.py > gpurun_out/${tag}_step_small_ops.txt 2>&1
ls -la gpurun_out
