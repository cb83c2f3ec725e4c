#!/bin/bash
# round 4, call 16: the wave-per-centre hierarchical-softmax kernel with the negatives under commit locks (k_sgns_train_hsw<.., NLOCK>): tests, bench, quality
set -o pipefail
O=gpurun_out/r04_run16; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |quality|Memory access|Error" $O/$name.log | tail -8; return $rc; }
run hs_tests 600 tests/test_gpu_sgns.py -k "hierarchical" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
echo "== cfg3 --hs default (negatives under locks)"; date; timeout -k 10 300 $B 2>$O/hs_locks.err | tee $O/hs_locks.json | line || exit 1
echo "== cfg3 --hs negatives by atomics"; timeout -k 10 300 $B --tune hs_centre=1 2>$O/hs_atomics.err | tee $O/hs_atomics.json | line
for w in 2304 3072; do echo "== cfg3 --hs locks workers=$w"; timeout -k 10 300 $B --tune workers=$w 2>$O/hs_locks_w$w.err | tee $O/hs_locks_w$w.json | line; done
for d in 8 16; do echo "== cfg3 --hs locks hs_drain=$d"; timeout -k 10 300 $B --tune hs_drain=$d 2>$O/hs_locks_d$d.err | tee $O/hs_locks_d$d.json | line; done
echo "== quality_scale hs"; date; timeout -k 10 500 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -5 $O/quality_hs.txt
date
