#!/bin/bash
# round 4, call 42 (A/B on one box): the wave-per-centre kernel asking for a round's context rows one round ahead
set -o pipefail
O=gpurun_out/r04_run42; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
timeout -k 10 300 python -m pytest tests/test_gpu_sgns.py -x -q -m gpu -k "linear_regime or hogwild_and_exchange" > $O/t.log 2>&1; tail -1 $O/t.log
ab() { echo "cfg3 --hs:"; $B 2>>$O/err | line; echo "cfg3 --hs again:"; $B 2>>$O/err | line; echo "cfg3_zipf --hs:"; $B --workload cfg3_zipf 2>>$O/err | line; }
echo "== new build"; ab
C=embedding_amd/csrc
cp $C/sgns_kernels.h $O/sgns_kernels.h.new; cp $C/sgns_kernels.h.orig $C/sgns_kernels.h
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
echo "== committed build"; ab
