#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of one launch of every kernel family of the bs-32 forward: where do writes / reads exceed the algorithmic bytes?
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; bash tools/traffic_pmc.sh tf_$name "$@" > gpurun_out/tf_$name.log 2>&1; find gpurun_out/tf_$name -name "*.db" -delete; }
run pw256_76   1 1 256 128 76 5 0
run pw512_38   1 1 512 256 38 5 0
run ig1024_19  1 1 1024 512 19 5 0
run ig64_128_152 3 1 64 128 152 5 0
run mp512_19   3 1 512 1024 19 5 0
run stem_s2    3 2 32 64 152 5 0
run l0         3 1 8 32 608 5 0
python3 - <<'PY' > gpurun_out/r05_traffic_families.txt
import json,glob,os
print("# HBM traffic per launch of the forward's kernel families at bs 32 (tools/traffic_pmc.sh: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes)")
print("%-26s %-34s %10s %10s %10s %8s" % ("kernel","shape","fetch MB","write MB","alg MB","ratio"))
for f in sorted(glob.glob("gpurun_out/tf_*/traffic.json")):
    d=json.load(open(f))
    if "hbm_bytes_per_launch" not in d: print(f, d.get("error")); continue
    print("%-26s %-34s %10.1f %10.1f %10.1f %8.3f" % (d["kernel"],d["shape"],d["fetch_bytes_per_launch"]/1e6,d["write_bytes_per_launch"]/1e6,d["algorithmic_bytes_per_launch"]/1e6,d["ratio_to_algorithmic"]))
PY
cat gpurun_out/r05_traffic_families.txt
