"""How large may the owner-computes mini-batch of an 8-rank block be?  dge_sorted_batch_items bounds it by 128 items per live row AND 2 048 terms for the BUSIEST row; on
cfg3 the second bound binds (8.8 M items: the busiest vertex holds 29x the mean count) although the average row then takes 70 items.  cfg3-sized community graph, one epoch
in 10 global batches of 1 M walks, 8 ranks simulated on one device (serial), link AUC and loss on held-out steps against the one-GPU run — with the mini-batch at its default
and forced to 1/2 and the whole of an episode's walks (DGE_TUNE_SORTED_WALKS).   python scripts/blocks_minibatch_quality.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from embedding_amd import synth
from helpers import link_auc_device, simulate_block_schedule, simulate_gather_syn0

R, T, L, D, K, N = 41667, 24, 24, 128, 5, 8
NV = R * T; dev = "cuda:0"
G = synth.flow_graph_torch(R, T, 100, dev, dst="community")
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); del G
torch.cuda.empty_cache(); g.build_alias(False)
epoch = 10 * NV
corpus = g.sample_walks_device(epoch, L, seed=20171106)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
test = torch.from_numpy(g.sample_walks(100_000, L, seed=99, rng_mode=1)).to(dev).to(torch.int64)
c = counts[counts >= 2].double(); print("rows %d, busiest row %.0f x the mean count" % (len(c), float(c.max() / c.mean())), flush=True)
cfg = E.make_config(D, L, NV, negative=K, workers=0, epochs=1, seed=1)
nb = epoch // 10

def words_of(lo, n):
    sub = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, sub, lo, n)
    return int(sub[counts >= 2].sum().item())

one = E.SgnsModel.create(cfg, counts, 0); vid = one.vectors()[1]
wb = 0
for b in range(10):
    one.train(corpus, b * nb, nb, walk_index_base=b * nb, words_before=wb, total_walks=epoch); wb += words_of(b * nb, nb)
print("one GPU, 10 batches                         AUC %.4f loss %.4f" % link_auc_device(one, vid, test, R, NV), one.schedule(), flush=True)
one.close()
for name, knob in (("default (4 mini-batches an episode)", {}), ("2 mini-batches an episode", {"sorted_walks": nb // 2 + 1}), ("1 mini-batch an episode", {"sorted_walks": nb})):
    with E.tuning(**knob):
        ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(N)]
        wb = 0; t0 = time.time()
        for b in range(10):
            lo = b * nb
            simulate_block_schedule(ms, lambda m: m.train(corpus, lo, nb, walk_index_base=lo, words_before=wb, total_walks=epoch), serial=True)
            wb += words_of(lo, nb)
        simulate_gather_syn0(ms)
        ker = sum(m.stats()["kernel_ms"] for m in ms)
    print("8 ranks, %-36s AUC %.4f loss %.4f" % ((name,) + link_auc_device(ms[0], vid, test, R, NV)), ms[0].schedule(), "kernel time of all ranks %.1f s, wall %.0f s" % (ker / 1e3, time.time() - t0), flush=True)
    for m in ms:
        m.close()
