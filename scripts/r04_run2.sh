#!/bin/bash
# round 4, second GPU call: the new tests again (ranks of the one-device simulation serialised; quality test resized), each in its own process
set -o pipefail
O=gpurun_out/r04_run2; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; date; timeout -k 10 $1 python -m pytest "${@:2}" -x -q -s -m gpu --durations=5 > $O/$name.log 2>&1; rc=$?; echo "rc $rc" >> $O/$name.log; grep -E "passed|failed|error|rc |quality|Memory access" $O/$name.log | tail -8; return $rc; }
run quality 600 tests/test_gpu_quality.py
run blocks_cfg3 600 tests/test_gpu_blocks_scale.py
run cfg5_tenth 400 tests/test_gpu_configs.py -k "tenth" && run cfg5_full8 600 tests/test_gpu_configs.py -k "cfg5_full_size_eight or cfg5_at_full"
run fit_search 300 tests/test_gpu_sgns.py -k "one_shot_fit or vec_writer"
echo "== epoch bench"; date
timeout -k 10 600 python bench.py --epoch --cpu-seconds 10 2>$O/epoch.err | tee $O/epoch.json | cut -c1-1500
date
