# the README table's other rows with the final build
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  frac %.3f  %.1f ms/launch  sched %s  placement search %s' % (d['value'], r['frac'], r['ms_per_launch'], r['schedule'], d['config'].get('placement_search')), flush=True)"; }
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 2>/dev/null | f cfg5
python3 bench.py --no-cpu-baseline --steps 2 --hs 2>/dev/null | f "cfg3 --hs"
python3 bench.py --no-cpu-baseline --steps 3 --workload cfg3_zipf 2>/dev/null | f cfg3_zipf
python3 bench.py --no-cpu-baseline --steps 5 --workload cfg2 2>/dev/null | f cfg2
python3 bench.py --no-cpu-baseline --steps 5 2>/dev/null | f cfg3
