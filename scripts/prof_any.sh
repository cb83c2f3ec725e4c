#!/bin/bash
# scripts/prof_any.sh <tag> <python script> [args...]: rocprofv3 kernel + HIP API stats of one python command -> gpurun_out/stats_<tag>.csv, hip_<tag>.csv
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/st_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
script=$1; shift
rocprofv3 --kernel-trace --hip-runtime-trace --stats --output-format csv -d $out -o s -- python3 $root/$script "$@" > $out/log.txt 2>&1
cp $(find $out -name "*kernel_stats.csv" | head -1) $root/gpurun_out/stats_$tag.csv
cp $(find $out -name "*hip_api_stats.csv" | head -1) $root/gpurun_out/hip_$tag.csv
rm -rf $out
