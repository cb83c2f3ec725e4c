#!/bin/bash
# round 4, seventh GPU call: auto-policy sweep (full) and the counter profiles of the headline and of cfg2
set -o pipefail
O=gpurun_out/r04_run7; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== policy sweep"; date
timeout -k 10 1000 python scripts/policy_sweep.py > $O/policy_sweep.txt 2>&1; tail -42 $O/policy_sweep.txt
echo "== profiles: cfg3 (headline)"; date
TRAFFIC_KEY=cfg3/policy5 TRAFFIC_X2="k_sgns_train_locked" timeout -k 10 600 bash scripts/collect_profiles.sh r04_cfg3 k_sgns_train_locked > $O/prof_cfg3.log 2>&1; tail -4 $O/prof_cfg3.log | cut -c1-500
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 600 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1; tail -4 $O/prof_cfg2.log | cut -c1-500
cp profiles/traffic.json $O/traffic.json
date
