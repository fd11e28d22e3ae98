#!/bin/bash
# tools/inchain.sh [layer]: kernel trace + three counter passes of tools/inchain.py (separate runs: gpurun refuses --pmc with API traces,
# and FETCH_SIZE / WRITE_SIZE do not fit one pass), then the per-phase table -> gpurun_out/inchain/summary.txt
set -u
layer=${1:-14}
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/inchain
rm -rf $out; mkdir -p $out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python $root/tools/inchain.py --layer $layer --out $out/inchain_phases.json > $out/trace.log 2>&1 || echo "trace pass failed"
for c in GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o t -- python $root/tools/inchain.py --layer $layer --out $out/phases_$c.json > $out/$c.log 2>&1 || echo "$c pass failed"
done
cd $root
python tools/inchain_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
# keep the merged-back scratch small: the CSVs are several MB each
find $out -name "*.csv" -size +2M -delete
