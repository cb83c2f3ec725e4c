# sclk / power / temperature of the card while the default bench runs three times in a row (profiles/r02_box_drift.txt)
root=$(pwd)
( for i in $(seq 1 110); do
    echo "t=$i $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|fclk|mclk|Power|Temperature' | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';' | tr -s ' \t' ' ')"
    sleep 0.5
  done ) > gpurun_out/clock_watch.txt 2>&1 &
WATCH=$!
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['ms_per_launch'], d['roofline'].get('box_copy_GBps'))"; }
for r in 1 2 3; do
  echo "run $r starts $(date +%s.%N)" >> gpurun_out/clock_runs.txt
  python3 bench.py --no-cpu-baseline --steps 6 --warmup 1 2>/dev/null | f run$r
done
wait $WATCH
