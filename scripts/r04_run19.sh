#!/bin/bash
# round 4, call 19: cfg2 with one unit per emit group + timeline
set -o pipefail
O=$(pwd)/gpurun_out/r04_run19; mkdir -p $O
root=$(pwd)
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.3f ms/step  %.3f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 20 --warmup 3"
for i in 1 2; do echo "== cfg2"; timeout -k 10 300 $B --workload cfg2 2>$O/cfg2_$i.err | tee $O/cfg2_$i.json | line || exit 1; done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 --placement-candidates 1 --workload cfg2 > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
cd $root
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py $f 130 > $O/timeline.txt
rm -rf $O/trace
grep -E "emit|count|phase" $O/timeline.txt | tail -14
