#!/bin/bash
# scripts/prof_any.sh <tag> <python script> [args...]: rocprofv3 kernel + HIP API stats of one python command -> gpurun_out/stats_<tag>.csv, hip_<tag>.csv
tag=$1; shift
for a in "$@"; do case "$prev" in --gpus) if [ "$a" -gt 1 ] 2>/dev/null; then echo "$0: --gpus $a: profile one rank per rocprofv3 process (bench.py would start further processes under the profiler)" >&2; exit 2; fi;; esac; prev=$a; done
root=$(pwd); out=$root/gpurun_out/st_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
script=$1; shift
rocprofv3 --kernel-trace --hip-runtime-trace --stats --output-format csv -d $out -o s -- python3 $root/$script "$@" > $out/log.txt 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) $root/gpurun_out/stats_$tag.csv
cp $(find $out -name "*hip_api_stats.csv" | head -1) $root/gpurun_out/hip_$tag.csv
rm -rf $out
