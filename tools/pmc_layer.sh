#!/bin/bash
# two rocprofv3 --pmc passes (SQ wave-time breakdown + MFMA busy; GRBM / LDS) on one conv shape:  tools/pmc_layer.sh <out> k s cin cout hout tile
set -e
out=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p $root/gpurun_out/$out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace -d $root/gpurun_out/$out/a -o a -- python $root/tools/one_layer.py "$@" > $root/gpurun_out/$out/a.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS \
  --kernel-trace -d $root/gpurun_out/$out/b -o b -- python $root/tools/one_layer.py "$@" > $root/gpurun_out/$out/b.log 2>&1 || true
cd $root
for d in a b; do
  db=$(find gpurun_out/$out/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db --pmc | grep "PMC.*conv" > gpurun_out/$out/$d.txt || echo "no db for $d" > gpurun_out/$out/$d.txt
  tail -3 gpurun_out/$out/$d.log
done
cat gpurun_out/$out/a.txt gpurun_out/$out/b.txt
