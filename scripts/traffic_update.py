"""python scripts/traffic_update.py <workload/policyN> <gpurun_out/prof_TAG> [--x2 SUBSTR|SUBSTR...] [--per-launch]
Rewrites one entry of profiles/traffic.json from the counter passes scripts/collect_profiles.sh just made (pmc.csv + bench.json of the
profiled process) and stamps it with the source hash of the kernels the loaded library was built from (dge_build_stamp): bench.py quotes an
entry only while the stamp matches — a kernel change that is not re-profiled prints `traffic: null` instead of a stale figure.

Rules (profiles/README.md, MI355X_MICROARCH.md HBM section): bytes = FETCH_SIZE x 1024 (x 2 for the kernels named by --x2: their 16-byte-per-lane
loads are tallied at half — checked against TCC_EA0_RDREQ x 128 B) + WRITE_SIZE x 1024; requests = TCC_EA0_RDREQ + TCC_EA0_WRREQ.  Every dispatch in
pmc.csv that matched collect_profiles.sh's kernel pattern is summed and divided by the launches of the counter run (bench.py --steps 2 --warmup 1 =
3 launches of the step; a kernel that runs several times per step — the owner-computes schedule — is summed per step)."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    key, prof = sys.argv[1], sys.argv[2]
    x2 = sys.argv[sys.argv.index("--x2") + 1].split("|") if "--x2" in sys.argv else []
    launches = 3
    bench = json.loads(open(os.path.join(prof, "bench.json")).read())
    pairs = bench["roofline"]["pairs_per_launch"] * (bench["roofline"].get("launches_per_step", 1))
    tot = {}; names = set()
    for r in csv.DictReader(open(os.path.join(prof, "pmc.csv"))):
        v = float(r["value"])
        if r["counter"] == "FETCH_SIZE" and any(s in r["kernel"] for s in x2):
            v *= 2.0
        tot[r["counter"]] = tot.get(r["counter"], 0.0) + v
        names.add(r["kernel"].replace("void ", "").split("(")[0])
    per = {k: v / launches / pairs for k, v in tot.items()}
    fetch = per.get("FETCH_SIZE", 0.0) * 1024.0; write = per.get("WRITE_SIZE", 0.0) * 1024.0
    import embedding_amd as E
    stamp = E.engine.build_stamp()["sorted" if key.endswith("policy8") else "kernels"]
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path))
    table[key] = {
        "kernel": " + ".join(sorted(names))[:400],
        "bytes_per_pair": round(fetch + write, 1), "fetch_bytes_per_pair": round(fetch, 1), "write_bytes_per_pair": round(write, 1),
        "requests_per_pair": round(per.get("TCC_EA0_RDREQ_sum", 0.0) + per.get("TCC_EA0_WRREQ_sum", 0.0), 2),
        "atomic_requests_per_pair": round(per.get("TCC_EA0_ATOMIC_sum", 0.0), 2),
        "stamp": stamp,
        "source": "scripts/traffic_update.py from %s (pmc.csv: %d dispatches summed over %d launches of %.4e pairs; FETCH_SIZE x1024%s + WRITE_SIZE x1024; "
                  "TCC_EA0_RDREQ %.2f + WRREQ %.2f requests per pair); ms_per_launch of the profiled process %.2f"
                  % (os.path.basename(prof.rstrip("/")), sum(1 for _ in open(os.path.join(prof, "pmc.csv"))) - 1, launches, pairs,
                     " (x2 for %s)" % "|".join(x2) if x2 else "", per.get("TCC_EA0_RDREQ_sum", 0.0), per.get("TCC_EA0_WRREQ_sum", 0.0), bench["roofline"]["ms_per_launch"]),
    }
    if key == "cfg3/policy5":
        table["cfg3/policy0"] = dict(table[key])
    json.dump(table, open(path, "w"), indent=1)
    print(key, json.dumps(table[key])[:600])


if __name__ == "__main__":
    main()
