#!/bin/bash
# round 4, call 17: cfg2's launch as a timeline (which kernels, which gaps)
set -o pipefail
O=$(pwd)/gpurun_out/r04_run17; mkdir -p $O
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 --placement-candidates 1 --workload cfg2 > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
cd $root
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py $f 120 > $O/timeline.txt
rm -rf $O/trace
tail -70 $O/timeline.txt
grep '^{"metric"' $O/bench.log | cut -c1-200
