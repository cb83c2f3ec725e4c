#!/bin/bash
# The lock kernel on the uniform cfg3 graph at other row widths and negative counts (one process per line):  bash scripts/shape_sweep.sh
for shape in "64 5" "128 5" "128 20" "192 5" "256 5" "256 20" "384 5" "512 5"; do
  set -- $shape
  python bench.py --workload cfg3 --dim $1 --negative $2 --no-cpu-baseline --steps 2 --warmup 1 --placement-candidates 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('D=$1 K=$2  %.3e edges/s  frac %.3f  %.1f ms/launch  rows read %.0f / read+written back %.0f GB/s  workers %s  tables %s' % (d['value'], r['frac'], r['ms_per_launch'], r['row_read_GBps'], r['row_rewrite_GBps'], r['schedule']['workers'], d['config'].get('table_placement')))"
done
