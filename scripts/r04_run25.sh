#!/bin/bash
# round 4, call 25: the stamped profiles of the final build (cfg3 headline, cfg3 --hs, cfg2) and the driver's bench command
set -o pipefail
O=gpurun_out/r04_run25; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tail -1 $O/build.log
echo "== profiles: cfg3"; date
TRAFFIC_KEY=cfg3/policy5 TRAFFIC_X2="k_sgns_train_locked" timeout -k 10 500 bash scripts/collect_profiles.sh r04_cfg3 k_sgns_train_locked > $O/prof_cfg3.log 2>&1 || { tail -5 $O/prof_cfg3.log; exit 1; }; tail -1 $O/prof_cfg3.log | cut -c1-200
python3 scripts/traffic_update.py cfg3/policy0 gpurun_out/prof_r04_cfg3 --x2 "k_sgns_train_locked" > /dev/null 2>&1
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 400 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1 || { tail -5 $O/prof_cfg2.log; exit 1; }; tail -1 $O/prof_cfg2.log | cut -c1-200
echo "== profiles: hs"; date
TRAFFIC_KEY=cfg3/hs TRAFFIC_X2="k_sgns_train_hsw" timeout -k 10 800 bash scripts/collect_profiles.sh r04_hs k_sgns_train_hsw --hs > $O/prof_hs.log 2>&1 || { tail -5 $O/prof_hs.log; exit 1; }; tail -1 $O/prof_hs.log | cut -c1-200
cp profiles/traffic.json $O/traffic.json
echo "== the driver's command"; date
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | tee $O/bench_default.json | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_run25/bench_default.json').read())
r=d['roofline']; print('value %.4e  frac %.3f  ms/step %.1f  ms/launch %.1f  traffic %s  cpu %s' % (d['value'], r['frac'], d['ms_per_step'], r['ms_per_launch'], r['traffic'], d['cpu_baseline']['value']))
PY
date
