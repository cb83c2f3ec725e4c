#!/bin/bash
# round 4: cfg5 in 8-rank blocks — fewer workers / smaller heads (experiment)
set -o pipefail
O=gpurun_out/r04_run15; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  %.1f ms/launch  sched %s' % (d['value'], r['ms_per_launch'], r['schedule']), flush=True)"; }
B="python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 --sim-ranks 8 --placement-candidates 1"
$B 2>/dev/null | f "auto"
for w in 3072 4608; do for h in 15000 30000 60320; do
  $B --tune workers=$w --tune hot_rows=$h 2>/dev/null | f "workers $w head $h"
done; done
date
