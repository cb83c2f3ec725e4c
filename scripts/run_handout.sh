f() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1   %.3e frac %.3f ms/launch %.2f  sched %s'%(d['value'], r['frac'], r['ms_per_launch'], r['schedule']))"; }
for t in "" "--tune static_walks=1"; do
python bench.py --no-cpu-baseline --steps 3 $t 2>/dev/null | f "cfg3 $t"
python bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf $t 2>/dev/null | f "cfg3_zipf $t"
python bench.py --no-cpu-baseline --steps 2 --sim-ranks 2 $t 2>/dev/null | f "sim2 $t"
python bench.py --no-cpu-baseline --steps 2 --workload cfg5 $t 2>/dev/null | f "cfg5 $t"
done
