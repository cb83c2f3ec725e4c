# does rocprofv3 slow the kernel, or does the box drift?  plain, plain, profiled, plain — the bench's own HIP-event ms per launch each time
root=$(pwd); export TMPDIR=/tmp
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['ms_per_launch'], d['roofline'].get('box_copy_GBps'))"; }
python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | f plain1
python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | f plain2
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pvp -o s -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | grep '^{' | f profiled; cd $root
python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | f plain3
