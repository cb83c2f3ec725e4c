#!/bin/bash
# round 4, call 30: the tree term under the block schedule, one rank of 8 / of 2 (never measured at cfg3 size)
set -o pipefail
O=gpurun_out/r04_run30; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 1 --warmup 1 --hs"
echo "== cfg3 --hs --sim-ranks 8"; timeout -k 10 900 $B --sim-ranks 8 2>$O/hs8.err | tee $O/hs8.json | line || { tail -5 $O/hs8.err; exit 1; }
echo "== cfg3 --hs --sim-ranks 2"; timeout -k 10 600 $B --sim-ranks 2 2>$O/hs2.err | tee $O/hs2.json | line
date
