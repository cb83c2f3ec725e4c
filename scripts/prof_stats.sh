#!/bin/bash
# scripts/prof_stats.sh <tag> [bench.py args...]: rocprofv3 kernel stats of one bench command -> gpurun_out/stats_<tag>.csv
tag=$1; shift
for a in "$@"; do case "$prev" in --gpus) if [ "$a" -gt 1 ] 2>/dev/null; then echo "$0: --gpus $a: profile one rank per rocprofv3 process (bench.py would start further processes under the profiler)" >&2; exit 2; fi;; esac; prev=$a; done
root=$(pwd); out=$root/gpurun_out/st_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/log.txt 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) $root/gpurun_out/stats_$tag.csv
rm -rf $out
head -14 $root/gpurun_out/stats_$tag.csv | cut -c1-200
