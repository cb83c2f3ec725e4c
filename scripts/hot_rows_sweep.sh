# the head size of the mixed policy 7 swept around the count-derived default (profiles/r02_hot_rows_sweep.txt); usage: hot_rows_sweep.sh <workload> <rows>...
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']), flush=True)"; }
w=$1; shift
python3 bench.py --no-cpu-baseline --steps 2 --workload $w --placement-trials 1 2>/dev/null | f "$w auto"
for h in "$@"; do
python3 bench.py --no-cpu-baseline --steps 2 --workload $w --placement-trials 1 --tune hot_rows=$h 2>/dev/null | f "$w hot_rows=$h"
done
