#!/bin/bash
# round 4, thirteenth GPU call: the final build — whole suite, counters of cfg2 (packed items) and a check of cfg3's, the driver's own bench command
set -o pipefail
O=gpurun_out/r04_run13; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tail -2 $O/build.log
echo "== full gpu suite"; date
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "rc $?" >> $O/gpu_tests.log; tail -16 $O/gpu_tests.log | cut -c1-200
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 600 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1; tail -1 $O/prof_cfg2.log | cut -c1-200
cp profiles/traffic.json $O/traffic.json
echo "== the driver's command"; date
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | tee $O/bench_default.json | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_run13/bench_default.json').read())
r=d['roofline']; print('value %.4e  frac %.3f  ms/step %.1f  ms/launch %.1f  traffic %s  cpu %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['traffic'], d['cpu_baseline']['value']))
PY
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()"
date
