#!/bin/bash
# round 4, call 29: k_block_emit looping over the batch's walks with the table's run form: block-schedule parity tests, one rank of 8, its kernel statistics
set -o pipefail
O=$(pwd)/gpurun_out/r04_run29; mkdir -p $O
root=$(pwd)
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -8; return $rc; }
run sorted 600 tests/test_gpu_sorted.py tests/test_gpu_sgns.py -k "sorted or owner or block" || exit 1
run dist 300 tests/test_gpu_distributed.py || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.3f ms/step  %.3f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
for i in 1; do echo "== cfg3 sim 8"; timeout -k 10 400 python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --sim-ranks 8 2>$O/sim8.err | tee $O/sim8_$i.json | line || exit 1; done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --placement-candidates 1 --sim-ranks 8 > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
cd $root
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
grep -E "k_block|k_sorted_count" $O/kernel_stats.csv | cut -c1-60,80-200
date
