#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_run44
timeout -k 10 500 python scripts/head_row_rates.py cfg5 1.18e8 2>/dev/null | tee gpurun_out/r04_run44/cfg5.txt
timeout -k 10 300 python scripts/head_row_rates.py cfg3_zipf 6.8e8 2>/dev/null | tee gpurun_out/r04_run44/cfg3_zipf.txt
