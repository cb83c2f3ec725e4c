#!/bin/bash
# round 4, third GPU call: the wave-per-centre hierarchical-softmax kernel (tests + bench), the block schedule's quality against batch size and
# learning-rate rule, the statistical parity test again
set -o pipefail
O=gpurun_out/r04_run3; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |quality|Memory access|Error" $O/$name.log | tail -8; return $rc; }
run hs_tests 600 tests/test_gpu_sgns.py -k "hierarchical or block_schedule" || exit 1
run hs_dist 300 tests/test_gpu_distributed.py -k "oracles_block_run"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
echo "== cfg3 --hs wave per centre"; date; timeout -k 10 300 $B 2>$O/hs_centre.err | tee $O/hs_centre.json | line
echo "== cfg3 --hs pair by pair (round 3)"; timeout -k 10 300 $B --tune hs_centre=0 2>$O/hs_pair.err | tee $O/hs_pair.json | line
for d in 2 8 16; do echo "== cfg3 --hs wave per centre hs_drain=$d"; timeout -k 10 300 $B --tune hs_drain=$d 2>$O/hs_centre_d$d.err | tee $O/hs_centre_d$d.json | line; done
echo "== cfg1 --hs wave per centre"; timeout -k 10 300 $B --workload cfg1 2>$O/hs1_centre.err | tee $O/hs1_centre.json | line
echo "== cfg1 --hs pair by pair"; timeout -k 10 300 $B --workload cfg1 --tune hs_centre=0 2>$O/hs1_pair.err | tee $O/hs1_pair.json | line
echo "== quality_scale hs"; date; timeout -k 10 500 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -5 $O/quality_hs.txt
run quality 600 tests/test_gpu_quality.py
echo "== blocks quality experiment"; date; timeout -k 10 900 python scripts/r04_blocks_quality.py > $O/blocks_quality.txt 2>&1; tail -30 $O/blocks_quality.txt
date
