#!/bin/bash
# round 4, call 40: atomics wave reading all its flags in one look + the tree's busiest nodes in copies — tests, then A/B against the committed build on one box
set -o pipefail
O=gpurun_out/r04_run40; mkdir -p $O
cd "$(dirname "$0")/.."
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |Memory access|Error" $O/$name.log | tail -8; return $rc; }
run tests 800 tests/test_gpu_sgns.py tests/test_gpu_policy.py -k "hierarchical or atomics or mixed or hot or policy or locked" || exit 1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1"
ab() {
echo "cfg3 --hs:"; $B --hs 2>>$O/err | line
echo "cfg3 --hs, LDS accumulators 15 KB drain 8:"; $B --hs --tune hs_hot_kb=15 --tune hs_drain=8 2>>$O/err | line
echo "cfg3_zipf --hs:"; $B --hs --workload cfg3_zipf 2>>$O/err | line
echo "cfg3_zipf:"; $B --workload cfg3_zipf 2>>$O/err | line
echo "cfg5:"; python bench.py --no-cpu-baseline --placement-candidates 1 --steps 2 --warmup 1 --workload cfg5 2>>$O/err | line
}
echo "== new build"; ab
C=embedding_amd/csrc
for f in sgns_kernels.h sgns.hip sgns_model.h; do cp $C/$f $O/$f.new; cp $C/$f.orig $C/$f; done
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
echo "== committed build"; ab
