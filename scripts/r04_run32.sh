#!/bin/bash
# round 4, call 32: SQ counters of the owner-computes phase kernels on cfg2 (are they issue bound?)
set -o pipefail
O=$(pwd)/gpurun_out/r04_run32; mkdir -p $O
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc -o p$i -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 --placement-candidates 1 --workload cfg2 > $O/pmc_$i.log 2>&1 || { tail -5 $O/pmc_$i.log; exit 1; }
done
cd $root
mkdir -p $O/flat; find $O/pmc -name "*.csv" -exec cp {} $O/flat/ \;
python3 scripts/pmc_digest.py $O/flat "k_sorted_phase|k_sorted_emit" > $O/sq.csv
rm -rf $O/pmc $O/flat
python3 - <<'PY'
import csv,collections
acc=collections.defaultdict(lambda: [0.0,0])
for r in csv.DictReader(open('gpurun_out/r04_run32/sq.csv')):
    k=(r['kernel'].split('(')[0].replace('void ',''), r['counter'])
    acc[k][0]+=float(r['value']); acc[k][1]+=1
for (kern,ctr),(v,n) in sorted(acc.items()):
    print('%-34s %-22s %.4e per dispatch (%d dispatches)' % (kern, ctr, v/n, n))
PY
