# every workload's bench line with the round's build on ONE box -> gpurun_out/final_numbers.txt (copied to profiles/rNN_final_numbers.txt)
out=gpurun_out/final_numbers.txt; : > $out
python3 -c "import embedding_amd as E; print('[build] libdge.so ABI v%d, kernel sources %s' % (E.lib.dge_version(), E.lib.dge_build_stamp().decode()))" >> $out
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$1  %.3e edges/s  frac %.3f  %.2f ms/launch  %s  sched %s  batch %s walks%s' % (d['value'], r['frac'], r['ms_per_launch'], r['kernel'].split(' (')[0], r['schedule'], c['global_batch_walks'], ('  lock_stats %s' % c['lock_stats']) if c.get('lock_stats') and c['lock_stats']['rounds'] else ''), flush=True)" >> $out; }
python3 bench.py --no-cpu-baseline --steps 5 2>/dev/null | f cfg3
python3 bench.py --no-cpu-baseline --steps 3 --workload cfg3_zipf 2>/dev/null | f cfg3_zipf
python3 bench.py --no-cpu-baseline --steps 5 --workload cfg2 2>/dev/null | f cfg2
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 2>/dev/null | f cfg5
python3 bench.py --no-cpu-baseline --steps 3 --workload cfg1 2>/dev/null | f cfg1
python3 bench.py --no-cpu-baseline --steps 2 --hs 2>/dev/null | f "cfg3 --hs"
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --hs 2>/dev/null | f "cfg3_zipf --hs"
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 --hs 2>/dev/null | f "cfg5 --hs"
python3 bench.py --no-cpu-baseline --steps 3 --workload cfg1 --hs 2>/dev/null | f "cfg1 --hs"
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --sim-ranks $n 2>/dev/null | f "cfg3 --sim-ranks $n (global batch = epoch/10)"; done
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --steps 2 --sim-ranks $n --weak-batch 2>/dev/null | f "cfg3 --sim-ranks $n --weak-batch"; done
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg5 --sim-ranks 8 2>/dev/null | f "cfg5 --sim-ranks 8"
python3 bench.py --no-cpu-baseline --steps 2 --workload cfg3_zipf --sim-ranks 8 2>/dev/null | f "cfg3_zipf --sim-ranks 8"
python3 bench.py --no-cpu-baseline --epoch 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py --epoch  epoch %.2f s  %.3e edges per wall-clock s  stages %s' % (d['epoch_s'], d['value'], d['stages_s']))" >> $out
python3 bench.py --no-cpu-baseline --epoch --hs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py --epoch --hs  epoch %.2f s  %.3e edges per wall-clock s  stages %s' % (d['epoch_s'], d['value'], d['stages_s']))" >> $out
date >> $out; cat $out
