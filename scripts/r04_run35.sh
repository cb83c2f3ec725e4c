#!/bin/bash
# round 4, call 35 (experiment): the owner-computes phases at higher occupancy (launch bounds 8 / 5 waves a SIMD for 64- / 128-float rows, a few spills), A/B on one box
set -o pipefail
O=gpurun_out/r04_run35; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.3f ms/launch' % (d['value'], r['frac'], r['ms_per_launch']))
"; }
ab() {
  for i in 1 2; do echo "cfg2:"; python bench.py --no-cpu-baseline --placement-candidates 1 --steps 20 --warmup 3 --workload cfg2 2>/dev/null | line; done
  echo "sim8:"; python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --sim-ranks 8 2>/dev/null | line
}
echo "== experiment build (as pushed)"; ab
cp embedding_amd/csrc/sgns_sorted.hip $O/exp.hip; cp embedding_amd/csrc/sgns_sorted.hip.orig embedding_amd/csrc/sgns_sorted.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
echo "== original"; ab
cp $O/exp.hip embedding_amd/csrc/sgns_sorted.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build2.log 2>&1
echo "== experiment again"; ab
