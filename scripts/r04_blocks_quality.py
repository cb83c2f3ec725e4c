"""Why does the 8-rank block schedule lose AUC when the global batch is most of an epoch?  (round 4; tests/test_gpu_blocks_scale.py found 0.915 against the
one-GPU 0.959 with 8 M walks as ONE global batch; with 10 batches per epoch round 3 measured 0.9593 against 0.9611.)
cfg3-sized community graph, 8 000 064 walks, one-GPU against 8 simulated ranks:
  - learning-rate horizon 1 epoch (alpha decays over the 10 M-walk epoch) and a constant alpha (horizon 1000 epochs);
  - the 8 M walks as 1, 4, 8 global batches;
  - lock kernels in the blocks (no item store) with the learning rate of a pair taken by WALK (as shipped) and by PROGRESS (episode e of a batch trains at
    position words_before + (e * batch_words + words_before_walk) / N)."""
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
TOT = N * (epoch // 10)

def words_of(lo, n):
    sub = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, sub, lo, n)
    return int(sub[counts >= 2].sum().item())

for horizon in (1, 1000):
    for pol in (0, 5):
        cfg = E.make_config(D, L, NV, negative=K, workers=0, epochs=horizon, seed=1, update_policy=pol)
        if pol == 0:
            one = E.SgnsModel.create(cfg, counts, 0); vid = one.vectors()[1]
            one.train(corpus, 0, TOT, walk_index_base=0, total_walks=epoch)
            print("horizon %4d  one GPU                                  AUC %.4f loss %.4f" % ((horizon,) + link_auc_device(one, vid, test, R, NV)), one.schedule(), flush=True)
            one.close()
        for nb in (1, 4, 8):
            for rule in (("walk",) if pol == 0 else ("walk", "progress")):
                ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(N)]
                wb = 0; t0 = time.time()
                for b in range(nb):
                    lo, n = b * (TOT // nb), TOT // nb
                    bw = words_of(lo, n)
                    if rule == "walk":
                        fn = lambda m: m.train(corpus, lo, n, walk_index_base=lo, words_before=wb, total_walks=epoch)
                    else:
                        fn = lambda m, e: m.train(corpus, lo, n, walk_index_base=lo, words_before=wb + e * bw // N, words_scale=1.0 / N, total_walks=epoch)
                    simulate_block_schedule(ms, fn, serial=True)
                    wb += bw
                simulate_gather_syn0(ms)
                print("horizon %4d  8 ranks, policy %d, %d batch(es), lr by %-8s AUC %.4f loss %.4f" % ((horizon, pol, nb, rule) + link_auc_device(ms[0], vid, test, R, NV)),
                      ms[0].schedule(), "%.0f s" % (time.time() - t0), flush=True)
                for m in ms:
                    m.close()
