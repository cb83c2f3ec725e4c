# owner-computes after a change: parity first, then the two workloads that use it
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_sorted.py -x -q 2>&1 | grep -v amdgpu.ids | tail -2
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.3e edges/s  %.2f ms/launch  trials %s' % (d['value'], r['ms_per_launch'], d['config']['placement_trial_ms']), flush=True)"; }
python3 bench.py --no-cpu-baseline --steps 5 --workload cfg2 2>/dev/null | f cfg2
python3 bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 2>/dev/null | f sim8
python3 bench.py --no-cpu-baseline --steps 2 --sim-ranks 4 2>/dev/null | f sim4
