#!/bin/bash
set -o pipefail
O=gpurun_out/r04_run34; mkdir -p $O
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_sorted.py -x -q -s -m gpu --durations=5 > $O/sorted.log 2>&1; rc=$?; echo "rc $rc" >> $O/sorted.log; grep -E "passed|failed|error|rc |Error|assert" $O/sorted.log | cut -c1-400 | tail -12
