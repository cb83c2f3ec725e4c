#!/bin/bash
# round 4, call 23: seven-wave workgroups of the hierarchical-softmax kernel: accumulator count x drain period, speed and quality
set -o pipefail
O=gpurun_out/r04_run23; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
for v in "hs_hot_kb=30 hs_drain=8" "hs_hot_kb=30 hs_drain=6" "hs_hot_kb=15 hs_drain=8" "hs_hot_kb=60 hs_drain=6"; do
  t=""; for kv in $v; do t="$t --tune $kv"; done
  echo "== cfg3 --hs seven waves $v"; timeout -k 10 300 $B $t 2>>$O/bench.err | line || exit 1
done
echo "== quality"; date
DGE_HS_VARIANTS="hs_centre=3,hs_hot_kb=30,hs_drain=8;hs_centre=3,hs_hot_kb=30,hs_drain=6;hs_centre=3,hs_hot_kb=15,hs_drain=8;hs_centre=2,hs_drain=8" timeout -k 10 700 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -5 $O/quality_hs.txt
date
