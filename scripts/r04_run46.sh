#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_run46
timeout -k 10 600 python -m pytest tests/test_gpu_sgns.py -x -q -m gpu -k "linear_regime" --durations=3 2>&1 | tail -6
