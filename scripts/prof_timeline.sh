#!/bin/bash
# scripts/prof_timeline.sh <tag> <n_dispatches> [bench.py args...]: rocprofv3 kernel trace of one bench command ->
#   gpurun_out/<tag>_kernel_stats.csv (per-kernel totals) and gpurun_out/<tag>_timeline.txt (the last n dispatches: start, duration, gap, queue)
tag=$1; n=$2; shift 2
for a in "$@"; do case "$prev" in --gpus) if [ "$a" -gt 1 ] 2>/dev/null; then echo "$0: --gpus $a: profile one rank per rocprofv3 process" >&2; exit 2; fi;; esac; prev=$a; done
root=$(pwd); out=$root/gpurun_out/tl_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/log.txt 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) $root/gpurun_out/${tag}_kernel_stats.csv
python3 $root/scripts/trace_timeline.py $(find $out -name "*kernel_trace.csv" | head -1) $n > $root/gpurun_out/${tag}_timeline.txt
grep '^{"metric"' $out/log.txt | tail -1 > $root/gpurun_out/${tag}_bench.json
rm -rf $out
head -16 $root/gpurun_out/${tag}_kernel_stats.csv | cut -c1-160; tail -3 $root/gpurun_out/${tag}_timeline.txt
