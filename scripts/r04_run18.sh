#!/bin/bash
# round 4, call 18: owner-computes emit with two units a group: parity tests + cfg2 / block-schedule bench
set -o pipefail
O=gpurun_out/r04_run18; mkdir -p $O
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -8; return $rc; }
run sorted_tests 600 tests/test_gpu_sorted.py tests/test_gpu_configs.py::test_cfg2_shaped_auto_schedule_resolves_to_owner_computes || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.3f ms/step  %.3f ms/launch  sched %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 20 --warmup 3"
for i in 1 2; do echo "== cfg2"; timeout -k 10 300 $B --workload cfg2 2>$O/cfg2_$i.err | tee $O/cfg2_$i.json | line || exit 1; done
for t in 0; do echo "== cfg2 table_runs=0"; timeout -k 10 300 $B --workload cfg2 --tune table_runs=0 2>$O/cfg2_noruns.err | tee $O/cfg2_noruns.json | line; done
date
