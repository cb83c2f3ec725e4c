"""python scripts/stats_table.py <kernel_stats.csv> [rows]: rocprofv3 kernel statistics as a short table (name, calls, total ms, average us, share)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    n = r["Name"]
    n = n if len(n) < 100 else n[:60] + " ... " + n[-30:]
    print("%-100s calls %6s total %8.1f ms avg %9.1f us %6s%%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
