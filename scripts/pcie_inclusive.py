"""The PCIe-inclusive rate of the hot path: bench.py's `value` is quoted with the walks resident in HBM; a host that hands the
library HOST buffers (dge_walks_from_host / dge_train_sgns, include/dge.h — what a JNI caller with a Java int[] does) pays the host link on top.

One cfg3 batch (1 000 008 walks x 24 tokens, 96 MB of int32; 3.83e8 pairs), the same model and kernel as `python bench.py`:
  resident   corpus already on the device                          -> the bench line's figure
  from host  dge_walks_from_host (pageable numpy memory) + train    -> PCIe-inclusive, per batch
  vectors    dge_model_vectors: syn0 to the host (512 MB), once per fit
  one shot   dge_train_sgns(host walks): count, vocabulary, unigram table, weights, train — the reference's w2v.fit() on a host corpus

Usage (GPU box): python scripts/pcie_inclusive.py > gpurun_out/pcie_inclusive.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import embedding_amd as E
    from embedding_amd import synth
    dev = "cuda:0"
    R, T, L, D, K = 41667, 24, 24, 128, 5
    NV = R * T
    G = synth.flow_graph_torch(R, T, 100, dev)
    g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); del G
    torch.cuda.empty_cache()
    g.build_alias(exact=False)
    B = NV
    epoch_walks = 10 * NV
    corpus = g.sample_walks_device(B, L, seed=20171106, rng_mode=1, first_index=0)
    counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
    cfg = E.make_config(D, L, NV, negative=K, min_count=2, epochs=1000, workers=0, seed=1)
    model = E.SgnsModel.create(cfg, counts, 0)
    walks_h = corpus.to_host()
    print("batch: %d walks x %d tokens = %.1f MB of int32 on the host" % (B, L, walks_h.nbytes / 1e6), flush=True)

    def timed(fn, n=3):
        out = []
        for _ in range(n):
            model.stats(); torch.cuda.synchronize()
            t = time.perf_counter(); fn(); model.stats(); torch.cuda.synchronize()
            out.append(time.perf_counter() - t)
        return out

    kw = dict(walk_index_base=0, epoch=0, words_before=0, words_scale=1.0, total_walks=epoch_walks)
    model.train(corpus, 0, B, **kw); model.stats()                                   # warm-up (work buffers, clocks)
    model.reset_stats()
    res = timed(lambda: model.train(corpus, 0, B, **kw))
    pairs = model.stats()["pairs"] / len(res)
    print("resident : %s ms per batch -> %.3e edges/s (%.0f pairs a batch; kernel %s)" % (", ".join("%.1f" % (x * 1e3) for x in res), pairs / min(res), pairs, model.kernel()
          if hasattr(model, "kernel") else "?"), flush=True)

    def from_host():
        c = E.WalkCorpus.from_host(walks_h, 0)
        model.train(c, 0, B, **kw)
        model.stats()
        c.close()
    res_h = timed(from_host)
    print("from host: %s ms per batch -> %.3e edges/s  (PCIe-inclusive: + %.1f ms = %.1f %% of a batch; the copy alone moves %.1f GB/s)" % (
        ", ".join("%.1f" % (x * 1e3) for x in res_h), pairs / min(res_h), (min(res_h) - min(res)) * 1e3, 100 * (min(res_h) - min(res)) / min(res),
        walks_h.nbytes / max(min(res_h) - min(res), 1e-9) / 1e9), flush=True)

    t = time.perf_counter(); s0, vid = model.vectors(); dt = time.perf_counter() - t
    print("vectors  : syn0 %d x %d (%.0f MB) to the host in %.1f ms (%.1f GB/s incl. the numpy copy) — once per fit" % (s0.shape[0], s0.shape[1], s0.nbytes / 1e6, dt * 1e3, s0.nbytes / dt / 1e9), flush=True)
    model.close(); corpus.close()

    cfg1 = E.make_config(D, L, NV, negative=K, min_count=2, epochs=1, workers=0, seed=1)
    for _ in range(2):
        t = time.perf_counter(); m = E.SgnsModel.fit(walks_h, cfg1, 0); st = m.stats(); dt = time.perf_counter() - t
        print("one shot : dge_train_sgns on the host batch: %.1f ms wall (vocabulary, table, weights, copy and train; kernel %.1f ms) -> %.3e edges/s end to end" % (
            dt * 1e3, st["kernel_ms"], st["pairs"] / dt), flush=True)
        m.close()


if __name__ == "__main__":
    main()
