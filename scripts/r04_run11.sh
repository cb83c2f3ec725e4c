#!/bin/bash
# round 4, eleventh GPU call: counters and kernel statistics of the owner-computes schedule with packed items (cfg2; one rank of 8)
set -o pipefail
O=gpurun_out/r04_run11; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.2f ms/step  %.2f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1"
echo "== cfg3 --sim-ranks 2 (10-bit digits)"; timeout -k 10 300 $B --sim-ranks 2 2>$O/sim2.err | tee $O/sim2.json | line
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 600 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1; tail -3 $O/prof_cfg2.log | cut -c1-300
python scripts/stats_table.py gpurun_out/prof_r04_cfg2/kernel_stats.csv 14
cp profiles/traffic.json $O/traffic.json
echo "== kernel statistics: one rank of 8"; date
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/sim8 -o s -- python3 $R/bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 --placement-candidates 1 > $R/$O/sim8_stats.log 2>&1
cd $R; cp $(find $O/sim8 -name "*kernel_stats.csv" | head -1) $O/sim8_kernel_stats.csv; rm -rf $O/sim8
python scripts/stats_table.py $O/sim8_kernel_stats.csv 14
grep '^{"metric"' $O/sim8_stats.log | line
date
