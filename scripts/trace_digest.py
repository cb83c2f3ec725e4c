"""python scripts/trace_digest.py <..._kernel_trace.csv> <kernel substring> <n>: duration statistics of the LAST n dispatches of a kernel in a rocprofv3
kernel trace (the timed region of bench.py: its warm-up and steps come last; the placement search's probe launches of the same kernel come before)."""
import csv, sys
path, needle, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))
     if any(x in r["Kernel_Name"] for x in needle.split("|"))]
d.sort()
last = d[-n:]
if not last:
    sys.exit("no dispatch of %s in %s" % (needle, path))
ms = [x[1] / 1e6 for x in last]
print("%s: %d dispatches in the trace; the last %d: average %.3f ms, min %.3f, max %.3f  (%s)" % (needle, len(d), len(last), sum(ms) / len(ms), min(ms), max(ms), last[-1][2][:80]))
