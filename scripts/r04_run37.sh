#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_run37
timeout -k 10 300 scripts/micro/hot_row_spread.bin | tee gpurun_out/r04_run37/hot_row_spread.txt
