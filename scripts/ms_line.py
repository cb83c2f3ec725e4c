"""stdin: one bench.py JSON line -> 'value frac ms/launch' (helper of scripts/ab.sh)"""
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("%.3e edges/s  frac %.3f  %.2f ms/launch  policy %s" % (d["value"], r["frac"], r["ms_per_launch"], r["schedule"]["update_policy"]))
