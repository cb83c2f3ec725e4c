#!/bin/bash
# round 4, call 31: the atomics form of the wave-per-centre kernel in seven-wave workgroups (skewed vocabularies)
set -o pipefail
O=gpurun_out/r04_run31; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 2 --warmup 1 --hs --workload cfg3_zipf"
echo "== cfg3_zipf --hs auto (atomics, three waves)"; timeout -k 10 400 $B 2>>$O/bench.err | line || exit 1
echo "== cfg3_zipf --hs atomics, seven waves"; timeout -k 10 400 $B --tune hs_centre=4 2>>$O/bench.err | line
echo "== cfg3_zipf --hs atomics, seven waves, 30 KB"; timeout -k 10 400 $B --tune hs_centre=4 --tune hs_hot_kb=30 2>>$O/bench.err | line
echo "== cfg3_zipf --hs atomics, seven waves, drain 4"; timeout -k 10 400 $B --tune hs_centre=4 --tune hs_drain=4 2>>$O/bench.err | line
echo "== cfg3 --hs atomics, seven waves"; timeout -k 10 400 python bench.py --no-cpu-baseline --placement-candidates 1 --steps 2 --warmup 1 --hs --tune hs_centre=4 2>>$O/bench.err | line
date
