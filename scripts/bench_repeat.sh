#!/bin/bash
# scripts/bench_repeat.sh <n> [bench.py args...]: the default bench n times back to back, one process each (consecutive processes are where the
# placement luck shows: profiles/r02_box_drift.txt) -> one line per run: edges/s, roofline fraction, ms per launch, the placement search's report
n=$1; shift
for i in $(seq 1 $n); do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2> gpurun_out/bench_repeat_$i.err | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); r=o['roofline']; c=o['config']
        print('run $i: %.3e edges/s  frac %.3f  %.1f ms/launch  step %.1f ms  tables %s  search %s  copy %.0f GB/s' % (o['value'], r['frac'], r['ms_per_launch'], o['ms_per_step'], c.get('table_placement'), c.get('placement_search'), r['box_copy_GBps'] or 0), flush=True)
"
done
