#!/usr/bin/env python3
"""Timeline of the LAST `n` dispatches of a rocprofv3 kernel trace: start offset, duration, gap to the previous dispatch's end — to see what a launch of a
multi-kernel schedule (owner-computes: ~16 kernels a mini-batch) spends between its kernels.  scripts/trace_timeline.py <kernel_trace.csv> <n>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:100]))
    busy += e - s; prev_end = max(prev_end, e)
print("span %.1f us, kernels %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
