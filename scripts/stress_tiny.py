"""Tail-latency stress of the commit-lock kernel on a 3-row vocabulary (extreme contention): repeats and prints the slowest runs."""
import sys, time, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np, embedding_amd as E
    workers, pol, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(0)
    walks = rng.integers(0, 3, (400, 5)).astype(np.int32)
    corpus = E.WalkCorpus.from_host(walks, 0)
    c = E.make_config(8, 5, 3, min_count=1, workers=workers, table_size=101, update_policy=pol)
    dm = E.SgnsModel.fit(walks, c, 0)
    ts = []
    for r in range(reps):
        t = time.perf_counter(); dm.train(corpus); dm.stats(); ts.append(time.perf_counter() - t)
    ts = sorted(ts)
    print("workers", workers, "pol", pol, "reps", reps, "median %.4fs max %.4fs" % (ts[len(ts) // 2], ts[-1]), flush=True)
else:
    for workers in (16, 64, 400):
        for pol in (5, 6):
            try:
                r = subprocess.run([sys.executable, __file__, str(workers), str(pol), "300"], capture_output=True, text=True, timeout=120)
                print(r.stdout.strip()[-120:], flush=True)
            except subprocess.TimeoutExpired:
                print(workers, pol, "TIMEOUT", flush=True)
