#!/bin/bash
# one rank of 8 on a skewed workload: the atomics wave's accumulator banks (rows a bank x updates a flush)
# usage: bash scripts/r05_block_acc.sh <workload> "<rows list>" "<drain list>"
wl=${1:-cfg3_zipf}; out=gpurun_out/r05_block_acc_$wl.txt; : > $out
line() {
  echo "== $wl --sim-ranks 8 $*" >> $out
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $wl --sim-ranks 8 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('%.3e edges/s  %.2f ms/launch  policy %s  head %s  workers %s  lock_stats %s' % (d['value'], r['ms_per_launch'], r['schedule']['update_policy'], r['schedule']['hot_rows'], r['schedule']['workers'], c['lock_stats']))" >> $out
}
line --tune acc_rows=0
for r in ${2:-16}; do for d in ${3:-2 4 8 16 64}; do line --tune acc_rows=$r --tune acc_drain=$d; done; done
cat $out
