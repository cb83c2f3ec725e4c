#!/bin/bash
# round 4, twelfth GPU call: the sorted launch with one host round trip
set -o pipefail
O=gpurun_out/r04_run12; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -6; return $rc; }
run sorted 600 tests/test_gpu_sorted.py tests/test_gpu_configs.py -k "owner or cfg2" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.2f ms/step  %.2f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 8 --warmup 2"
for i in 1 2; do echo "== cfg2"; timeout -k 10 300 $B --workload cfg2 2>$O/cfg2.err | tee $O/cfg2.json | line; done
echo "== cfg2 (300 000 walks a launch)"; timeout -k 10 300 $B --workload cfg2 --batch-walks 300000 2>$O/cfg2b.err | tee $O/cfg2b.json | line
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1"
echo "== cfg3 --sim-ranks 8, global batch 8 M"; timeout -k 10 300 $B --sim-ranks 8 2>$O/sim8.err | tee $O/sim8.json | line
echo "== cfg3 --sim-ranks 8, global batch 2 M"; timeout -k 10 300 $B --sim-ranks 8 --batch-walks 250002 2>$O/sim8_2m.err | tee $O/sim8_2m.json | line
echo "== cfg3 --sim-ranks 8, global batch 1 M"; timeout -k 10 300 $B --sim-ranks 8 --batch-walks 125001 2>$O/sim8_1m.err | tee $O/sim8_1m.json | line
echo "== cfg3 --sim-ranks 4"; timeout -k 10 300 $B --sim-ranks 4 2>$O/sim4.err | tee $O/sim4.json | line
echo "== cfg3 --sim-ranks 2"; timeout -k 10 300 $B --sim-ranks 2 2>$O/sim2.err | tee $O/sim2.json | line
echo "== cfg3 one GPU"; timeout -k 10 300 $B 2>$O/one.err | tee $O/one.json | line
date
