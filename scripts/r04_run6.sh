#!/bin/bash
# round 4, sixth GPU call: auto-policy sweep (full), counter profiles (cfg3 headline, cfg3 --hs), the epoch line
set -o pipefail
O=gpurun_out/r04_run6; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== policy sweep"; date
timeout -k 10 900 python scripts/policy_sweep.py > $O/policy_sweep.txt 2>&1; tail -42 $O/policy_sweep.txt
echo "== epoch"; date
timeout -k 10 600 python bench.py --epoch --cpu-seconds 10 2>$O/epoch.err | tee $O/epoch.json | cut -c1-900
timeout -k 10 600 python bench.py --epoch --hs --no-cpu-baseline 2>$O/epoch_hs.err | tee $O/epoch_hs.json | cut -c1-900
echo "== profiles: cfg3 --hs"; date
TRAFFIC_KEY=cfg3/hs TRAFFIC_X2="k_sgns_train_hsw" timeout -k 10 900 bash scripts/collect_profiles.sh r04_hs k_sgns_train_hsw --hs > $O/prof_hs.log 2>&1; tail -6 $O/prof_hs.log | cut -c1-600
date
