#!/bin/bash
# round 4, call 22: hierarchical softmax, a wave per centre in workgroups of seven training waves (k_sgns_train_hsw<.., NLOCK, 7>): tests, bench, quality
set -o pipefail
O=gpurun_out/r04_run22; mkdir -p $O
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |quality|Memory access|Error" $O/$name.log | tail -8; return $rc; }
run hs_tests 600 tests/test_gpu_sgns.py -k "hierarchical" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
echo "== cfg3 --hs default (locks, seven waves)"; date; timeout -k 10 300 $B 2>$O/hs_7.err | tee $O/hs_7.json | line || exit 1
echo "== cfg3 --hs locks, three waves"; timeout -k 10 300 $B --tune hs_centre=2 2>$O/hs_3.err | tee $O/hs_3.json | line
for d in 4 16; do echo "== cfg3 --hs seven waves hs_drain=$d"; timeout -k 10 300 $B --tune hs_drain=$d 2>$O/hs_7_d$d.err | tee $O/hs_7_d$d.json | line; done
echo "== quality_scale hs"; date; timeout -k 10 600 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -6 $O/quality_hs.txt
date
