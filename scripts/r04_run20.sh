#!/bin/bash
# round 4, call 20: what the block schedule's once-per-batch kernels take (k_block_count, k_block_emit) after the run-form / tally changes
set -o pipefail
O=$(pwd)/gpurun_out/r04_run20; mkdir -p $O
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --placement-candidates 1 --sim-ranks 8 > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
cd $root
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
head -14 $O/kernel_stats.csv | cut -c1-60,200-
grep -E "k_block|k_sorted_count" $O/kernel_stats.csv | cut -c1-200
grep '^{"metric"' $O/bench.log | cut -c1-160
