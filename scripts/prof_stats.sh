#!/bin/bash
# scripts/prof_stats.sh <tag> [bench.py args...]: rocprofv3 kernel stats of one bench command -> gpurun_out/stats_<tag>.csv
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/st_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/log.txt 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) $root/gpurun_out/stats_$tag.csv
rm -rf $out
head -14 $root/gpurun_out/stats_$tag.csv | cut -c1-200
