#!/bin/bash
# round 4, call 26: hierarchical softmax on the skewed workloads (never measured with the wave-per-centre kernel)
set -o pipefail
O=gpurun_out/r04_run26; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 2 --warmup 1 --hs"
echo "== cfg3_zipf --hs (auto)"; timeout -k 10 400 $B --workload cfg3_zipf 2>>$O/bench.err | line || exit 1
echo "== cfg3_zipf --hs pair by pair"; timeout -k 10 400 $B --workload cfg3_zipf --tune hs_centre=0 2>>$O/bench.err | line
echo "== cfg3_zipf --hs locks, seven waves (forced)"; timeout -k 10 400 $B --workload cfg3_zipf --tune hs_centre=3 2>>$O/bench.err | line
echo "== cfg5 --hs (auto: D = 256, pair by pair)"; timeout -k 10 600 $B --workload cfg5 --steps 1 2>>$O/bench.err | line
date
