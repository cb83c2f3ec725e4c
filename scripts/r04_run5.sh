#!/bin/bash
# round 4, fifth GPU call: the whole -m gpu suite (timing against the driver's 900 s), the auto-policy sweep, counter profiles of the new HS kernel
set -o pipefail
O=gpurun_out/r04_run5; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== full gpu suite"; date
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=25 -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "rc $?" >> $O/gpu_tests.log; tail -45 $O/gpu_tests.log | cut -c1-300
echo "== policy sweep (quick)"; date
timeout -k 10 600 python scripts/policy_sweep.py quick > $O/policy_sweep_quick.txt 2>&1; tail -12 $O/policy_sweep_quick.txt
date
