#!/bin/bash
# round 4, call 43: how many copies the root needs now that the atomics wave is out of the way (DGE_TUNE_HS_DRAIN = copies of the root), speed and quality
set -o pipefail
O=gpurun_out/r04_run43; mkdir -p $O
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%.3e edges/s  frac %.3f  %.1f ms/launch' % (d['value'], r['frac'], r['ms_per_launch']))
"; }
B="python bench.py --no-cpu-baseline --placement-candidates 1 --steps 3 --warmup 1 --hs"
for f in 32 24 16; do echo "root copies $f:"; $B --tune hs_drain=$f 2>>$O/err | line; done
echo "root copies 32 again:"; $B 2>>$O/err | line
timeout -k 10 300 python -m pytest tests/test_gpu_sgns.py -x -q -m gpu -k "linear_regime or lds_combining" > $O/t.log 2>&1; tail -1 $O/t.log
DGE_HS_VARIANTS="hs_centre=3" timeout -k 10 600 python scripts/quality_scale.py hs > $O/q.txt 2>&1; grep -v amdgpu.ids $O/q.txt | tail -2
