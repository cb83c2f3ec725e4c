#!/bin/bash
# one rank's share of an 8-rank step on the skewed workloads, with the block kernels' lock statistics (bench line: config.lock_stats), at the batch the product uses
# (a tenth of the epoch) and at the round-4 step (8 x that); then the one-GPU line of the same workload on the same box
out=gpurun_out/r05_block_schedule_skewed.txt; : > $out
for wl in cfg3_zipf cfg5; do
  for extra in "" "--weak-batch"; do
    echo "== $wl --sim-ranks 8 $extra" >> $out
    python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $wl --sim-ranks 8 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('%.3e edges/s  frac %.3f  %.2f ms/launch  policy %s  head %s  workers %s  global batch %d walks  lock_stats %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']['update_policy'], r['schedule']['hot_rows'], r['schedule']['workers'], c['global_batch_walks'], c['lock_stats']))" >> $out
  done
  echo "== $wl one GPU" >> $out
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $wl 2>/dev/null | python scripts/ms_line.py >> $out
done
cat $out
