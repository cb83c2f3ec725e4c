#!/bin/bash
# round 5: every committed profile of the round in one go, with the round's final build (the stamps of profiles/traffic.json follow the loaded library)
set -x
TRAFFIC_KEY=cfg3/policy5 TRAFFIC_X2="k_sgns_train_locked" bash scripts/collect_profiles.sh r05_cfg3 k_sgns_train_locked > /dev/null 2>&1
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_finish" bash scripts/collect_profiles.sh r05_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > /dev/null 2>&1
TRAFFIC_KEY=cfg3/hs_centre TRAFFIC_X2="k_sgns_train_hsw" bash scripts/collect_profiles.sh r05_hs k_sgns_train_hsw --hs > /dev/null 2>&1
TRAFFIC_KEY=cfg3_zipf/policy7 TRAFFIC_X2="k_sgns_train_locked" bash scripts/collect_profiles.sh r05_cfg3_zipf k_sgns_train_locked --workload cfg3_zipf > /dev/null 2>&1
TRAFFIC_KEY=cfg5/policy7 TRAFFIC_X2="k_sgns_train_locked" bash scripts/collect_profiles.sh r05_cfg5 k_sgns_train_locked --workload cfg5 > /dev/null 2>&1
TRAFFIC_KEY=cfg1/policy2 bash scripts/collect_profiles.sh r05_cfg1 k_sgns_train_small --workload cfg1 > /dev/null 2>&1
cp profiles/traffic.json gpurun_out/traffic.json
bash scripts/collect_profiles.sh r05_zipf_sim8 k_sgns_train_locked --workload cfg3_zipf --sim-ranks 8 > /dev/null 2>&1
bash scripts/collect_profiles.sh r05_cfg5_sim8 k_sgns_train_locked --workload cfg5 --sim-ranks 8 > /dev/null 2>&1
bash scripts/prof_timeline.sh r05_sim8_1M 120 --steps 3 --warmup 1 --sim-ranks 8 > /dev/null 2>&1
bash scripts/prof_timeline.sh r05_sim8 120 --steps 2 --warmup 1 --sim-ranks 8 --weak-batch > /dev/null 2>&1
for t in r05_cfg3 r05_cfg2 r05_hs r05_cfg3_zipf r05_cfg5 r05_cfg1 r05_zipf_sim8 r05_cfg5_sim8; do echo == $t; cat gpurun_out/prof_$t/kernel_timed.txt; cut -c1-300 gpurun_out/prof_$t/bench.json; echo; done
python3 -c "
import json; d=json.load(open('gpurun_out/traffic.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v.get('stamp'), v.get('bytes_per_pair'), v.get('requests_per_pair'), v.get('atomic_requests_per_pair'))"
