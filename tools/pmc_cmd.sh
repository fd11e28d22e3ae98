#!/bin/bash
# SQ counter passes on one command (separate rocprofv3 --pmc runs, --kernel-trace only: gpurun refuses --pmc with API traces):
#   tools/pmc_cmd.sh <out-dir-under-gpurun_out> <kernel-name-regex> <command...>
# pass a: wave-time buckets + MFMA busy, b: LDS, c: instruction mix / lane occupancy, d: MFMA instruction counts + fifo stalls.
# Prints "PMC <kernel> <counter> n avg sum" rows of the kernels matching the regex.
out=$1; pat=$2; shift 2
export TMPDIR=/tmp
root=$(pwd)
mkdir -p $root/gpurun_out/$out
cd /tmp
pass() {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $root/gpurun_out/$out/$name -o $name -- "${CMD[@]}" > $root/gpurun_out/$out/$name.log 2>&1 || echo "pass $name failed / timed out"
}
CMD=("$@")
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass b GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT
pass c SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
pass d SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL
cd $root
for d in a b c d; do
  db=$(find gpurun_out/$out/$d -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db --pmc | grep -E "^PMC" | grep -E "$pat" > gpurun_out/$out/$d.txt; else echo "no db for pass $d" > gpurun_out/$out/$d.txt; tail -5 gpurun_out/$out/$d.log; fi
  cat gpurun_out/$out/$d.txt
  rm -rf gpurun_out/$out/$d      # the raw rocpd databases are tens of MB; gpurun merges at most 64 MiB back
done
