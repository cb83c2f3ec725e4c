#!/bin/bash
# round 4, fourth GPU call
set -o pipefail
O=gpurun_out/r04_run4; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -6; grep "^\[quality" $O/$name.log | cut -c1-1800; return $rc; }
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
run hs_tests 600 tests/test_gpu_sgns.py -k "hierarchical"
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
for d in 4 6 8; do echo "== cfg3 --hs wave per centre hs_drain=$d"; timeout -k 10 300 $B --tune hs_drain=$d 2>$O/hs_centre_d$d.err | tee $O/hs_centre_d$d.json | line; done
run quality 600 tests/test_gpu_quality.py
run blocks_cfg3 600 tests/test_gpu_blocks_scale.py
run cfg5_tenth 400 tests/test_gpu_configs.py -k "tenth" && run cfg5_full8 600 tests/test_gpu_configs.py -k "cfg5_full_size_eight or cfg5_at_full"
B2="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1"
echo "== cfg3 --sim-ranks 8, global batch 8 M walks (weak scaling, the bench's default)"; timeout -k 10 300 $B2 --sim-ranks 8 2>$O/sim8.err | tee $O/sim8.json | line
echo "== cfg3 --sim-ranks 8, global batch 2 M walks"; timeout -k 10 300 $B2 --sim-ranks 8 --batch-walks 250002 2>$O/sim8_2m.err | tee $O/sim8_2m.json | line
echo "== cfg3 --sim-ranks 8, global batch 1 M walks"; timeout -k 10 300 $B2 --sim-ranks 8 --batch-walks 125001 2>$O/sim8_1m.err | tee $O/sim8_1m.json | line
echo "== quality_scale hs (drain sweep)"; date; DGE_HS_DRAINS="6,8" timeout -k 10 400 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -5 $O/quality_hs.txt
date
