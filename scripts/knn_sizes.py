"""dge_knn_cosine at sizes that fill the device evenly and unevenly (profiles/r02_knn_sizes.txt)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from embedding_amd import evaluate as ev

rng = np.random.default_rng(0)
for n, D in [(32768, 128), (41667, 128), (65536, 128), (98304, 128), (131072, 128), (41667, 64), (41667, 256)]:
    f = rng.normal(size=(n, D)).astype(np.float32)
    best = 1e9
    for _ in range(3):
        _, _, ms = ev.knn_cosine_gpu(f, 10)
        best = min(best, ms)
    print("n %6d D %3d: %7.2f ms  %5.1f TFLOP/s = %.0f %% of the f32 MFMA peak  (%.2f 16-row wave strips per SIMD)" % (n, D, best, 2.0 * n * n * D / best / 1e9, 2.0 * n * n * D / best / 1e9 / 1.57, (n + 15) // 16 / 1024.0), flush=True)
