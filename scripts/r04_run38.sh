#!/bin/bash
# round 4, call 38: the busiest inner nodes of the Huffman tree in copies instead of LDS accumulators (k_sgns_train_hsw, HS_REP): tests, bench, quality flat and skewed
set -o pipefail
O=gpurun_out/r04_run38; mkdir -p $O
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -8; return $rc; }
run hs_tests 600 tests/test_gpu_sgns.py -k "hierarchical" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
for rep in 1 2; do
echo "== cfg3 --hs copies, root 8 (default)"; timeout -k 10 300 $B 2>>$O/bench.err | line || exit 1
echo "== cfg3 --hs copies, root 4"; timeout -k 10 300 $B --tune hs_drain=4 2>>$O/bench.err | line
echo "== cfg3 --hs copies, root 2"; timeout -k 10 300 $B --tune hs_drain=2 2>>$O/bench.err | line
echo "== cfg3 --hs LDS accumulators 15 KB, drain 8 (before)"; timeout -k 10 300 $B --tune hs_hot_kb=15 --tune hs_drain=8 2>>$O/bench.err | line
done
echo "== cfg3_zipf --hs copies (atomics form)"; timeout -k 10 300 $B --workload cfg3_zipf 2>>$O/bench.err | line
echo "== cfg3_zipf --hs copies, root 4"; timeout -k 10 300 $B --workload cfg3_zipf --tune hs_drain=4 2>>$O/bench.err | line
echo "== cfg3_zipf --hs LDS accumulators (before)"; timeout -k 10 300 $B --workload cfg3_zipf --tune hs_hot_kb=30 2>>$O/bench.err | line
echo "== quality flat"; date
DGE_HS_VARIANTS="hs_centre=3;hs_centre=3,hs_drain=4;hs_centre=3,hs_hot_kb=15,hs_drain=8" timeout -k 10 700 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; grep -v amdgpu.ids $O/quality_hs.txt | tail -3
echo "== quality zipf"; date
DGE_HS_VARIANTS="hs_centre=1;hs_centre=1,hs_drain=4;hs_centre=1,hs_hot_kb=30" timeout -k 10 700 python scripts/quality_scale.py hs zipf > $O/quality_hs_zipf.txt 2>&1; grep -v amdgpu.ids $O/quality_hs_zipf.txt | tail -3
date
