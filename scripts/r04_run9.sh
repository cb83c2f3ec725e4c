#!/bin/bash
# round 4, ninth GPU call: the whole -m gpu suite after the table-placement bound, the sort microbenchmark, the epoch line again (pipelined .vec writer)
set -o pipefail
O=gpurun_out/r04_run9; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cat $O/build.log | tail -2
echo "== full gpu suite"; date
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=12 -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "rc $?" >> $O/gpu_tests.log; tail -22 $O/gpu_tests.log | cut -c1-300
echo "== sort microbenchmark"; date
timeout -k 10 120 scripts/micro/sort_keys64.bin > $O/sort_keys64.txt 2>&1; cat $O/sort_keys64.txt
echo "== epoch"; date
timeout -k 10 600 python bench.py --epoch --cpu-seconds 10 2>$O/epoch.err | tee $O/epoch.json | cut -c1-600
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()"
date
