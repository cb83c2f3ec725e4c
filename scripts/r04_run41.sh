#!/bin/bash
# round 4, call 41 (experiment): the wave-per-centre kernel with six training waves and TWO atomics waves a workgroup against seven and one
set -o pipefail
O=gpurun_out/r04_run41; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
timeout -k 10 300 python -m pytest tests/test_gpu_sgns.py -x -q -m gpu -k "linear_regime" > $O/t.log 2>&1; tail -1 $O/t.log
for i in 1 2; do
echo "7 + 1 (default):"; $B 2>>$O/err | line
echo "6 + 2:"; $B --tune hs_centre=5 2>>$O/err | line
done
echo "7 + 1, LDS accumulators:"; $B --tune hs_hot_kb=15 --tune hs_drain=8 2>>$O/err | line
echo "6 + 2, LDS accumulators:"; $B --tune hs_centre=5 --tune hs_hot_kb=15 --tune hs_drain=8 2>>$O/err | line
