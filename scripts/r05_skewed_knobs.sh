#!/bin/bash
# one rank of 8 on a skewed workload: the block's head size, the worker count and the syn0 rule swept one at a time around what the library picks
# usage: bash scripts/r05_skewed_knobs.sh <workload> [out]
wl=${1:-cfg3_zipf}; out=${2:-gpurun_out/r05_skewed_knobs_$wl.txt}; : > $out
line() {
  echo "== $wl --sim-ranks 8 $*" >> $out
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $wl --sim-ranks 8 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('%.3e edges/s  %.2f ms/launch  policy %s  head %s  workers %s  lock_stats %s' % (d['value'], r['ms_per_launch'], r['schedule']['update_policy'], r['schedule']['hot_rows'], r['schedule']['workers'], c['lock_stats']))" >> $out
}
line
for h in ${HEADS:-10000 20000 40000 160000 320000}; do line --tune hot_rows=$h; done
for w in ${WORKERS:-3072 4096 8192 12288}; do line --workers $w; done
line --tune block_syn0_free=1
cat $out
