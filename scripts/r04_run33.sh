#!/bin/bash
# round 4, call 33: the round's last build — whole GPU suite, smoke, cfg2's stamped counters again (sgns_sorted.hip changed since), the driver's bench command
set -o pipefail
O=gpurun_out/r04_run33; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tail -1 $O/build.log
echo "== full gpu suite"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_tests.log 2>&1; rc=$?; echo "rc $rc" >> $O/gpu_tests.log; tail -14 $O/gpu_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" || exit 1
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 400 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1 || { tail -5 $O/prof_cfg2.log; exit 1; }; tail -1 $O/prof_cfg2.log | cut -c1-200
cp profiles/traffic.json $O/traffic.json
echo "== the driver's command"; date
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | tee $O/bench_default.json | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_run33/bench_default.json').read())
r=d['roofline']; print('value %.4e  frac %.3f  ms/step %.1f  ms/launch %.1f  traffic %s  cpu %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['traffic'], d['cpu_baseline']['value']))
PY
timeout -k 10 300 python3 bench.py --workload cfg2 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg2 %.4e frac %.3f traffic %s' % (d['value'], r['frac'], r['traffic']))"
date
