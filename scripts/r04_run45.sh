#!/bin/bash
# round 4, call 45 (A/B on one box): the atomics wave's one-look flag sweep under the mixed block kernels (one rank of 8 on the skewed workloads)
set -o pipefail
O=gpurun_out/r04_run45; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  %.1f ms/launch' % (d['value'], r['ms_per_launch']))
"; }
ab() {
for i in 1 2; do echo "cfg3_zipf --sim-ranks 8:"; python bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --sim-ranks 8 --placement-candidates 1 2>>$O/err | line; done
echo "cfg5 --sim-ranks 8:"; python bench.py --no-cpu-baseline --steps 2 --workload cfg5 --sim-ranks 8 --placement-candidates 1 2>>$O/err | line
}
echo "== final build (one-look sweep)"; ab
C=embedding_amd/csrc
cp $C/sgns_kernels.h $O/k.new; cp $C/sgns_kernels.h.orig $C/sgns_kernels.h
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
echo "== the same with the box-by-box sweep"; ab
