"""Digest rocprofv3 --pmc passes (…_counter_collection.csv + …_kernel_trace.csv) into one small table per kernel of interest:
pass, dispatch, kernel, counter, value (summed over the rows rocprofv3 emits per dispatch), grid, workgroup, duration.
Usage: python scripts/pmc_digest.py <dir> <kernel-substring>[|<substring>...] > profiles/<name>.csv"""
import csv, glob, os, sys

d, needle = sys.argv[1], sys.argv[2]
print("pass_,dispatch,kernel,counter,value,grid,wg,dur_ns")
for cc in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    tag = os.path.basename(cc)[: -len("_counter_collection.csv")]
    dur = {}
    kt = os.path.join(d, tag + "_kernel_trace.csv")
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            dur[r.get("Dispatch_Id")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = {}
    for r in csv.DictReader(open(cc)):
        if not any(nd in r["Kernel_Name"] for nd in needle.split("|")):
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"], r["Grid_Size"], r["Workgroup_Size"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    for (disp, name, ctr, grid, wg), v in sorted(acc.items(), key=lambda kv: (kv[0][2], int(kv[0][0]))):
        print('%s,%s,"%s",%s,%s,%s,%s,%s' % (tag, disp, name, ctr, v, grid, wg, dur.get(disp, "")))
