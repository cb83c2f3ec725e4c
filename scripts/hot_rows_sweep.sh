f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  frac %.3f  %.1f ms/launch  sched %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule']), flush=True)"; }
for h in 1000 4000 30000 60000; do
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 --placement-trials 1 --tune hot_rows=$h 2>/dev/null | f "cfg5 hot_rows=$h"
done
for h in 5000 40000; do
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --placement-trials 1 --tune hot_rows=$h 2>/dev/null | f "zipf hot_rows=$h"
done
