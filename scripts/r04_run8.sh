#!/bin/bash
# round 4, eighth GPU call (the sweep with the moved constants, the suite, cfg2's counters): auto-policy sweep (full) and the counter profiles of the headline and of cfg2
set -o pipefail
O=gpurun_out/r04_run8; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== policy sweep"; date
timeout -k 10 1000 python scripts/policy_sweep.py > $O/policy_sweep.txt 2>&1; tail -42 $O/policy_sweep.txt
echo "== quality_scale hs (the shipped kernels)"; date; timeout -k 10 400 python scripts/quality_scale.py hs > $O/quality_hs.txt 2>&1; tail -3 $O/quality_hs.txt
echo "== full gpu suite"; date
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "rc $?" >> $O/gpu_tests.log; tail -22 $O/gpu_tests.log | cut -c1-300
echo "== profiles: cfg2"; date
TRAFFIC_KEY=cfg2/policy8 TRAFFIC_X2="k_sorted_phase|k_sorted_fixup|k_sorted_commit" timeout -k 10 600 bash scripts/collect_profiles.sh r04_cfg2 "k_sorted|rocprim|k_block" --workload cfg2 > $O/prof_cfg2.log 2>&1; tail -4 $O/prof_cfg2.log | cut -c1-500
cp profiles/traffic.json $O/traffic.json
date
