#!/bin/bash
# Same-box A/B of library builds: scripts/ab.sh <rounds> "<command>" <variant>...   (variants: build/libdge_<variant>.so)
# Processes on one box differ by +-5 %, so the variants are interleaved and repeated.
set -e
rounds=$1; cmd=$2; shift 2
cp embedding_amd/libdge.so build/libdge_keep.so
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    cp build/libdge_$v.so embedding_amd/libdge.so
    echo "== $v (round $i)"
    bash -c "$cmd" 2>&1 | grep -v "^\[\|amdgpu.ids" | tail -4
  done
done
cp build/libdge_keep.so embedding_amd/libdge.so
