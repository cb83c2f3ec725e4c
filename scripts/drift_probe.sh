# when does a box slow down, and does a pause bring it back?  (profiles/r02_box_drift.txt)
f() { python3 -c "import json,sys,time; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['ms_per_launch'],1), 'ms/launch', flush=True)"; }
for r in 1 2 3 4 5; do python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | f "run$r t=$(date +%s)"; done
echo "sleep 90"; sleep 90
for r in 6 7; do python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | f "run$r t=$(date +%s)"; done
echo "one long process: 40 steps"; python3 bench.py --no-cpu-baseline --steps 40 --warmup 1 2>/dev/null | f "long t=$(date +%s)"
