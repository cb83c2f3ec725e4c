#!/bin/bash
# round 4, tenth GPU call: the owner-computes schedule with packed 8-byte items — parity tests first, then the numbers
set -o pipefail
O=gpurun_out/r04_run10; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -6; grep "^\[" $O/$name.log | cut -c1-400; return $rc; }
run sorted 600 tests/test_gpu_sorted.py tests/test_gpu_fuzz.py -k "sorted or owner or word2vec_order or bit_exact" || exit 1
run blocks 600 tests/test_gpu_sgns.py tests/test_gpu_distributed.py tests/test_gpu_configs.py -k "block_schedule or two_rank or cfg2" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.2f ms/step  %.2f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 5 --warmup 2"
echo "== cfg2"; timeout -k 10 300 $B --workload cfg2 2>$O/cfg2.err | tee $O/cfg2.json | line
echo "== cfg2 (300 000 walks a launch)"; timeout -k 10 300 $B --workload cfg2 --batch-walks 300000 2>$O/cfg2b.err | tee $O/cfg2b.json | line
echo "== cfg3 --sim-ranks 8"; timeout -k 10 300 $B --steps 3 --warmup 1 --sim-ranks 8 2>$O/sim8.err | tee $O/sim8.json | line
echo "== cfg3 --sim-ranks 4"; timeout -k 10 300 $B --steps 3 --warmup 1 --sim-ranks 4 2>$O/sim4.err | tee $O/sim4.json | line
echo "== cfg3 --sim-ranks 2"; timeout -k 10 300 $B --steps 3 --warmup 1 --sim-ranks 2 2>$O/sim2.err | tee $O/sim2.json | line
run stats 900 tests/test_gpu_blocks_scale.py tests/test_gpu_quality.py tests/test_gpu_policy.py tests/test_gpu_configs.py -k "not cfg1 and not cfg5_shaped_graph"
date
