#!/bin/bash
# round 4, call 36: the tree term on the skewed community graph at full size (one epoch): pair by pair against the wave-per-centre kernel auto picks there (atomics form)
set -o pipefail
O=gpurun_out/r04_run36; mkdir -p $O
cd "$(dirname "$0")/.."
date
DGE_HS_VARIANTS="hs_centre=1,hs_drain=1;hs_centre=1,hs_cold=0;hs_centre=1,hs_drain=1,hs_cold=0;hs_centre=0,hs_drain=1" timeout -k 10 900 python scripts/quality_scale.py hs zipf > $O/quality_hs_zipf.txt 2>&1; grep -v amdgpu.ids $O/quality_hs_zipf.txt | tail -5
date
