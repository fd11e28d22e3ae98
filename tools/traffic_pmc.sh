#!/bin/bash
# HBM traffic of one conv shape: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the TCC block cannot hold both),
# 2 launches each.  tools/traffic_pmc.sh <out> k s cin cout hout tile     -> gpurun_out/<out>/{fetch,write}.txt
#                   tools/traffic_pmc.sh <out> wgrad k s cin cout hout    the weight gradient of that layer at bs 64 (tools/one_wgrad.py)
set -e
out=$1; shift
tool=one_layer.py; mode=""
if [ "$1" = wgrad ]; then tool=one_wgrad.py; mode=--wgrad; shift; fi
export TMPDIR=/tmp
root=$(pwd)
mkdir -p $root/gpurun_out/$out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace -d $root/gpurun_out/$out/$c -o t -- python $root/tools/$tool "$@" > $root/gpurun_out/$out/$c.log 2>&1 || echo "$c pass failed/timeout"
done
cd $root
python tools/traffic_summary.py gpurun_out/$out $mode "$@"
