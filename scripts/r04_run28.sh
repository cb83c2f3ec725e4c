#!/bin/bash
# round 4, call 28: the bench line of every workload with the round's last build on one box (profiles/r04_final_numbers.txt), the epoch end to end with and without
# the tree term, one rank of 8
set -o pipefail
O=gpurun_out/r04_run28; mkdir -p $O
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
tail -1 $O/build.log | tee $O/final_numbers.txt
bash scripts/run_final_numbers.sh 2>&1 | tee -a $O/final_numbers.txt
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']), flush=True)"; }
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg1 2>/dev/null | f cfg1 | tee -a $O/final_numbers.txt
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg1 --hs 2>/dev/null | f "cfg1 --hs" | tee -a $O/final_numbers.txt
python3 bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 --placement-candidates 1 2>/dev/null | f "cfg3 --sim-ranks 8" | tee -a $O/final_numbers.txt
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 --sim-ranks 8 --placement-candidates 1 2>/dev/null | f "cfg5 --sim-ranks 8" | tee -a $O/final_numbers.txt
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --sim-ranks 8 --placement-candidates 1 2>/dev/null | f "cfg3_zipf --sim-ranks 8" | tee -a $O/final_numbers.txt
e() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1  epoch %.2f s  stages %s  cpu projected %.0f s' % (d['epoch_s'], d['stages_s'], d['cpu_baseline']['projected_epoch_s']), flush=True)"; }
timeout -k 10 500 python3 bench.py --epoch 2>/dev/null | tee $O/epoch.json | e "bench.py --epoch" | tee -a $O/final_numbers.txt
timeout -k 10 500 python3 bench.py --epoch --hs --no-cpu-baseline 2>/dev/null | tee $O/epoch_hs.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py --epoch --hs  epoch %.2f s  stages %s' % (d['epoch_s'], d['stages_s']), flush=True)" | tee -a $O/final_numbers.txt
date
