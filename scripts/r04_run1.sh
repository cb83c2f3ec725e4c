#!/bin/bash
# round 4, first GPU call: the new driver-suite tests + the block schedule on skewed vocabularies (head / tail split inside a block)
set -o pipefail
O=gpurun_out/r04_run1; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== tests"; date
timeout -k 10 900 python -m pytest tests/test_gpu_quality.py tests/test_gpu_blocks_scale.py -x -q -s -m gpu --durations=10 > $O/tests_a.log 2>&1; echo "rc $?" >> $O/tests_a.log; tail -15 $O/tests_a.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -s -m gpu -k "cfg5" --durations=10 > $O/tests_b.log 2>&1; echo "rc $?" >> $O/tests_b.log; tail -8 $O/tests_b.log
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 2 --warmup 1"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/step  %.1f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
for wl in cfg3_zipf cfg5; do
  echo "== $wl one GPU"; date; timeout -k 10 300 $B --workload $wl 2>$O/${wl}_one.err | tee $O/${wl}_one.json | line
  echo "== $wl --sim-ranks 8 auto (head/tail split inside the block)"; timeout -k 10 300 $B --workload $wl --sim-ranks 8 2>$O/${wl}_sim8.err | tee $O/${wl}_sim8.json | line
  echo "== $wl --sim-ranks 8 --policy 2 (round 3: atomics on every row)"; timeout -k 10 300 $B --workload $wl --sim-ranks 8 --policy 2 2>$O/${wl}_sim8_p2.err | tee $O/${wl}_sim8_p2.json | line
  echo "== $wl --sim-ranks 8 syn0 never locked"; timeout -k 10 300 $B --workload $wl --sim-ranks 8 --tune block_syn0_free=1 2>$O/${wl}_sim8_free.err | tee $O/${wl}_sim8_free.json | line
  for h in 2000 8000 30000 120000; do
    echo "== $wl --sim-ranks 8 hot_rows=$h"; timeout -k 10 300 $B --workload $wl --sim-ranks 8 --tune hot_rows=$h 2>$O/${wl}_sim8_h$h.err | tee $O/${wl}_sim8_h$h.json | line
  done
done
date
