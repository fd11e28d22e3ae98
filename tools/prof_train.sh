# rocprofv3 kernel stats of a short train-step run; prints the rows matching $1 (default: loss|decode)
export TMPDIR=/tmp; root=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/p_t -o x -- python $root/bench.py --mode train --bs 64 --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-kernel-table > /dev/null 2> $root/gpurun_out/pt.err
cd $root
db=$(find gpurun_out/p_t -name "*_results.db" | head -1)
python tools/rocpd_summary.py $db > gpurun_out/pt_stats.txt
rm -rf gpurun_out/p_t
grep -iE "${1:-loss|decode}" gpurun_out/pt_stats.txt
