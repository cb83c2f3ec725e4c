"""Do the block kernels' accumulator banks cost the embedding anything?  In one block of an 8-rank schedule the partition's hottest rows of both tables add their
updates up in every workgroup's LDS and go out `drain` at a time (lk_atomics_wave, LkAcc): up to (workgroups x drain) of a row's updates are parked at any time.
cfg3-sized graph with Zipf-popular destinations ("zipf": popularity only; "community_zipf": communities of 64 + a fifth of the flow to popular regions), one epoch in 10
global batches of 1 M walks, 8 ranks simulated on one device (serial): link AUC and loss on held-out steps, the same for the pairs that END in one of the 192 busiest
vertices, and the mean norm of the 24 busiest rows — against the one-GPU run and against the banks switched off.
    python scripts/blocks_acc_quality.py [zipf|community_zipf]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import embedding_amd as E
from embedding_amd import synth
from helpers import link_auc_device, link_scores_device, simulate_block_schedule, simulate_gather_syn0

dst = sys.argv[1] if len(sys.argv) > 1 else "zipf"
R, T, L, D, K, N = 41667, 24, 24, 128, 5, 8
NV = R * T; dev = "cuda:0"
G = synth.flow_graph_torch(R, T, 100, dev, dst=dst)
g = E.DeviceGraph(0); g.add_edges_device(G["src"], G["dst"], G["w"]); g.set_sources(G["sources"]); del G
torch.cuda.empty_cache(); g.build_alias(False)
epoch = 10 * NV
corpus = g.sample_walks_device(epoch, L, seed=20171106)
counts = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, counts)
test = torch.from_numpy(g.sample_walks(100_000, L, seed=99, rng_mode=1)).to(dev).to(torch.int64)
c = counts[counts >= 2].double(); print("%s: rows %d, busiest row %.0f x the mean count" % (dst, len(c), float(c.max() / c.mean())), flush=True)
cfg = E.make_config(D, L, NV, negative=K, workers=0, epochs=1, seed=1)
nb = epoch // 10
busy = torch.topk(counts, 192).indices          # vertex ids of the 192 busiest


def words_of(lo, n):
    sub = torch.zeros(NV, dtype=torch.int64, device=dev); corpus.count_tokens(NV, sub, lo, n)
    return int(sub[counts >= 2].sum().item())


def report(name, m, vid, extra=""):
    auc, loss = link_auc_device(m, vid, test, R, NV)
    # held-out steps into the busiest vertices: the rows whose updates were parked
    steps = torch.stack([test[:, :-1].reshape(-1), test[:, 1:].reshape(-1)], 1)
    hot = torch.isin(steps[:, 1], busy) & (steps[:, 0] >= 0) & (steps[:, 1] >= 0)
    sub = test[hot.reshape(test.shape[0], L - 1).any(1)][:20000]
    auc_h, loss_h = link_auc_device(m, vid, sub, R, NV)
    syn0, _ = m.vectors()
    print("%-34s AUC %.4f loss %.4f | walks through the busiest 192: AUC %.4f loss %.4f | mean |syn0| of the 24 busiest rows %.3f, of all %.3f  %s" %
          (name, auc, loss, auc_h, loss_h, float(np.linalg.norm(syn0[:24], axis=1).mean()), float(np.linalg.norm(syn0, axis=1).mean()), extra), flush=True)


one = E.SgnsModel.create(cfg, counts, 0); vid = one.vectors()[1]
wb = 0
for b in range(10):
    one.train(corpus, b * nb, nb, walk_index_base=b * nb, words_before=wb, total_walks=epoch); wb += words_of(b * nb, nb)
report("one GPU, 10 batches", one, vid, str(one.schedule()))
one.close()
sets = (("8 ranks, banks off", {"acc_rows": 0}), ("8 ranks, 16 rows, 4 a flush", {"acc_rows": 16, "acc_drain": 4}), ("8 ranks, 16 rows, 8 a flush", {"acc_rows": 16, "acc_drain": 8}),
        ("8 ranks, 16 rows, 16 a flush", {"acc_rows": 16, "acc_drain": 16}), ("8 ranks, 16 rows, 64 a flush", {"acc_rows": 16, "acc_drain": 64}))
if "workers" in sys.argv[2:]:      # the library's banks, fewer workers a block: a block's rows take n times their one-GPU share of the pairs in flight
    sets = tuple(("8 ranks, default banks, %d workers" % w, {"workers": w}) for w in (6144, 4096, 3072, 2048))
if "head" in sys.argv[2:]:         # the block's head (rows that take atomics instead of commit locks): a block's speed does not depend on it (profiles/r05_skewed_knobs_*) — does the embedding?
    sets = (("8 ranks, the library's head", {}),) + tuple(("8 ranks, head %d rows" % h, {"hot_rows": h}) for h in (160000, 80000, 40000, 20000, 10000))
if "default" in sys.argv[2:]:      # what the library picks, nothing forced
    sets = (("8 ranks, library defaults", {}),)
for name, knob in sets:
    with E.tuning(**knob):
        ms = [E.SgnsModel.create(cfg, counts, 0) for _ in range(N)]
        wb = 0; t0 = time.time()
        for b in range(10):
            lo = b * nb
            simulate_block_schedule(ms, lambda m: m.train(corpus, lo, nb, walk_index_base=lo, words_before=wb, total_walks=epoch), serial=True)
            wb += words_of(lo, nb)
        simulate_gather_syn0(ms)
        ker = sum(m.stats()["kernel_ms"] for m in ms)
    report(name, ms[0], vid, "%s kernel time of all ranks %.1f s, wall %.0f s" % (ms[0].schedule(), ker / 1e3, time.time() - t0))
    for m in ms:
        m.close()
