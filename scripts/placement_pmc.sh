#!/bin/bash
# which hardware counter moves with a model's placement?  scripts/slow_state_probe.py (models A, B, C of one process, the same launch on each)
# under rocprofv3 --pmc, one counter set per process; per set: every k_sgns_train_locked dispatch with its duration and counter values
root=$(pwd); out=$root/gpurun_out/placement_pmc; rm -rf $out; mkdir -p $out/flat
export TMPDIR=/tmp; cd /tmp
i=0
sets=("$@")      # counter sets, one argument each ("A B"); none: the sets behind profiles/r02_box_drift.txt
if [ ${#sets[@]} -eq 0 ]; then sets=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"); fi
for ctr in "${sets[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/p$i -o set$i -- python3 $root/scripts/slow_state_probe.py > $out/log$i.txt 2>&1
  find $out/p$i -name "*.csv" -exec cp {} $out/flat/ \;
  grep pid $out/log$i.txt | sed -e "s/(rewrite [0-9]* GB\/s)//g; s/, walks w, w+W, ...: [0-9]* ms//g"
done
cd $root
python3 scripts/pmc_digest.py $out/flat "k_sgns_train_locked" > gpurun_out/placement_pmc.csv
rm -rf $out
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/placement_pmc.csv")))
by=collections.defaultdict(dict)
for r in rows: by[(r["pass_"], int(r["dispatch"]))][r["counter"]]=(float(r["value"]), int(r["dur_ns"] or 0))
for (p,d),v in sorted(by.items()):
    dur=list(v.values())[0][1]
    print(p, d, "%.1f ms" % (dur/1e6), " ".join("%s=%.4g" % (k, x[0]) for k,x in sorted(v.items())))
PY
