#!/bin/bash
# Collect the evidence behind one bench line on the GPU box:  scripts/collect_profiles.sh <tag> <kernel-substring> [bench.py args...]
#   gpurun_out/prof_<tag>/bench.json            the bench line printed BY THE PROFILED RUN (HIP-event timing, cpu baseline skipped): the same
#                                               process as kernel_stats.csv, so the two must agree
#   gpurun_out/prof_<tag>/bench_unprofiled.json the same command without the profiler, run second (consecutive processes on a box differ: profiles/r02_box_drift.txt)
#   gpurun_out/prof_<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof_<tag>/pmc.csv               FETCH_SIZE / WRITE_SIZE+TCC_EA0_ATOMIC / TCC_EA0_RDREQ+WRREQ, one pass each, digested per dispatch
#   gpurun_out/prof_<tag>/kernel_timed.txt      the same kernel trace restricted to the LAST steps + warm-up dispatches of the kernel: the placement search
#                                               (bench.py --placement-candidates, on by default) runs quarter-size probe launches of the same kernel
#                                               before the timed region, which rocprofv3's own average includes; this average is the one the bench
#                                               line's ms_per_launch must agree with
#   gpurun_out/prof_<tag>/traffic.json          (with TRAFFIC_KEY set) profiles/traffic.json with this workload's entry rewritten from pmc.csv and stamped
# Counters are collected in their own runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
# One process per rocprofv3: bench.py --gpus N > 1 starts further processes (a launcher hop the pool forbids under the profiler) — refused here;
# to profile several ranks export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT and run one rocprofv3 per rank.
set -e
tag=$1; needle=$2; shift 2
for a in "$@"; do case "$prev" in --gpus) if [ "$a" -gt 1 ] 2>/dev/null; then echo "collect_profiles.sh: --gpus $a: profile one rank per rocprofv3 process" >&2; exit 2; fi;; esac; prev=$a; done
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $root/scripts/trace_digest.py $(find $out/stats -name "*kernel_trace.csv" | head -1) "$needle" 5 > $out/kernel_timed.txt
grep '^{"metric"' $out/stats.log | tail -1 > $out/bench.json
python3 $root/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/bench_unprofiled.json 2> $out/bench.err
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc -o $name -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --placement-candidates 1 "$@" > $out/pmc_$name.log 2>&1
done
cd $root
mkdir -p $out/flat; find $out/pmc -name "*.csv" -exec cp {} $out/flat/ \;
python3 scripts/pmc_digest.py $out/flat "$needle" > $out/pmc.csv
rm -rf $out/stats $out/pmc $out/flat
# TRAFFIC_KEY=cfg3/policy5 [TRAFFIC_X2="k_sgns_train_locked"]: rewrite that entry of profiles/traffic.json from the passes just made, stamped with the build's
# kernel-source hash (scripts/traffic_update.py); the table travels back as $out/traffic.json (gpurun only merges gpurun_out/)
if [ -n "$TRAFFIC_KEY" ]; then python3 scripts/traffic_update.py "$TRAFFIC_KEY" $out ${TRAFFIC_X2:+--x2 "$TRAFFIC_X2"} && cp profiles/traffic.json $out/traffic.json; fi
head -3 $out/kernel_stats.csv; cat $out/kernel_timed.txt; cat $out/bench.json | cut -c1-400
