#!/bin/bash
# same-box A/B: round-4 tree (build_prev/r04, built from the round-4 sources) against this tree: one rank of 8 at the 1 M-walk global batch, and the one-GPU headline
out=gpurun_out/r05_ab_sim8.txt; : > $out
for rep in 1 2; do
  for tree in build_prev/r04 .; do
    echo "== $tree: --sim-ranks 8 --batch-walks 125001 (rep $rep)" >> $out
    (cd $tree && python bench.py --no-cpu-baseline --steps 6 --warmup 2 --sim-ranks 8 --batch-walks 125001 2>/dev/null) | python scripts/ms_line.py >> $out
  done
done
for tree in build_prev/r04 .; do
  echo "== $tree: one GPU, default" >> $out
  (cd $tree && python bench.py --no-cpu-baseline --steps 8 2>/dev/null) | python scripts/ms_line.py >> $out
done
cat $out
