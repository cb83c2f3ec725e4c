#!/bin/bash
# round 4, call 39 (A/B on one box): the wave-per-centre kernel as committed against the build with the busiest nodes in copies
set -o pipefail
O=gpurun_out/r04_run39; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch  box copy %s' % (d['value'], r['frac'], r['ms_per_launch'], r.get('box_copy_GBps')))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
echo "== new build: copies (default: root 8)"; $B 2>>$O/err | line
echo "== new build: LDS accumulators 15 KB drain 8"; $B --tune hs_hot_kb=15 --tune hs_drain=8 2>>$O/err | line
C=embedding_amd/csrc
for f in sgns_kernels.h sgns.hip sgns_model.h; do cp $C/$f $O/$f.new; cp $C/$f.orig $C/$f; done
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
echo "== committed build (LDS accumulators 15 KB drain 8)"; $B 2>>$O/err | line
