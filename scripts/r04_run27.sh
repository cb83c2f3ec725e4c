#!/bin/bash
# round 4, call 27: the statistical-parity tests with the hierarchical-softmax leg
set -o pipefail
O=gpurun_out/r04_run27; mkdir -p $O
cd "$(dirname "$0")/.."
date
timeout -k 10 900 python -m pytest tests/test_gpu_quality.py -x -q -s -m gpu --durations=5 > $O/quality.log 2>&1; rc=$?; echo "rc $rc" >> $O/quality.log
grep -E "passed|failed|error|rc |quality|Error|assert" $O/quality.log | cut -c1-1800 | tail -12
date
