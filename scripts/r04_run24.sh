#!/bin/bash
# round 4, call 24: whole GPU suite with the build of the round's last kernel changes (HS seven-wave workgroups, emit run form, count tallies, .vec formatter),
# the driver's bench command, smoke
set -o pipefail
O=gpurun_out/r04_run24; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tail -2 $O/build.log
echo "== full gpu suite"; date
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_tests.log 2>&1; rc=$?; echo "rc $rc" >> $O/gpu_tests.log; tail -16 $O/gpu_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" || exit 1
date
