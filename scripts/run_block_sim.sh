# one rank's share of an N-rank block-schedule step on one GPU (profiles/r02_block_schedule_sim.txt)
for n in 2 3 4 8; do
  echo "== --sim-ranks $n (auto)"; python bench.py --no-cpu-baseline --steps 2 --sim-ranks $n 2>/dev/null | python scripts/ms_line.py
done
echo "== --sim-ranks 8 --policy 7 (round-1 schedule: row locks on syn1neg, syn0 by atomics)"; python bench.py --no-cpu-baseline --steps 2 --sim-ranks 8 --policy 7 2>/dev/null | python scripts/ms_line.py
echo "== --sim-ranks 4 --policy 5 (round-1 schedule: row locks)"; python bench.py --no-cpu-baseline --steps 2 --sim-ranks 4 --policy 5 2>/dev/null | python scripts/ms_line.py
echo "== --sim-ranks 2 --policy 5 (row locks)"; python bench.py --no-cpu-baseline --steps 2 --sim-ranks 2 --policy 5 2>/dev/null | python scripts/ms_line.py
echo "== one GPU, default"; python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python scripts/ms_line.py
