#!/bin/bash
# round 4, call 21: the .vec writer's integer formatter (bytes against printf, epoch wall clock), block-schedule bench after the emit changes
set -o pipefail
O=gpurun_out/r04_run21; mkdir -p $O
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -8; return $rc; }
run vec 300 tests/test_gpu_sgns.py -k "vec_writer" || exit 1
run sorted 600 tests/test_gpu_sorted.py || exit 1
echo "== epoch"; timeout -k 10 600 python bench.py --epoch 2>$O/epoch.err | tee $O/epoch.json | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['epoch_s'], d['stages_s'])
"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.3f ms/step  %.3f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
echo "== cfg3 sim 8"; timeout -k 10 400 python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --sim-ranks 8 2>$O/sim8.err | tee $O/sim8.json | line
date
