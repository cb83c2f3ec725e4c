f() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1   %.3e frac %.3f ms/launch %.2f  sched %s'%(d['value'], r['frac'], r['ms_per_launch'], r['schedule']))"; }
python bench.py --workload cfg2 --no-cpu-baseline --steps 5 --policy 8 2>/dev/null | f "cfg2 pol8"
python bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 --policy 8 2>/dev/null | f "cfg3 sim8 pol8"
python bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 2>/dev/null | f "cfg3 sim8 auto"
python bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --policy 8 2>/dev/null | f "cfg3_zipf pol8"
python bench.py --no-cpu-baseline --steps 2 --workload cfg5 --policy 8 2>/dev/null | f "cfg5 pol8"
python bench.py --no-cpu-baseline --steps 2 --workload cfg5 2>/dev/null | f "cfg5 auto"
