"""GPU: seeded random sweep of the trainer's parameter space against the oracle — in-order workers, bit-exact.
Each case draws corpus shape (ragged walks, padding, out-of-vocabulary tokens), dimension, window, negatives, min_count,
epochs, table size, and the mode (plain / hierarchical softmax / multi-GPU block schedule).  Reference call site:
J/DeepWalk.java:73-79 (the SGNS half of the oracle is a restatement: parity unpinned)."""
import numpy as np
import pytest

from helpers import bits, simulate_block_schedule, simulate_gather_syn0

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    NV = int(rng.integers(8, 400))
    L = int(rng.choice([2, 3, 5, 8, 16, 17, 24, 40, 64]))
    n = int(rng.integers(20, 300))
    zipf = rng.random() < 0.5
    ids = (np.minimum(rng.zipf(1.4, size=(n, L)) - 1, NV - 1) if zipf else rng.integers(0, NV, (n, L))).astype(np.int32)
    lens = rng.integers(0, L + 1, n)                       # ragged: the tail of every walk is padding (-1)
    ids[np.arange(L)[None, :] >= lens[:, None]] = -1
    if rng.random() < 0.3:
        ids[rng.random(ids.shape) < 0.1] = -1              # holes inside walks too (dropped like out-of-vocabulary tokens)
    cfg = dict(dim=int(rng.choice([2, 8, 20, 33, 64, 65, 100, 128, 192, 200, 256, 384, 500])), window=int(rng.integers(1, L + 3)),
               negative=int(rng.choice([0, 1, 2, 5, 13, 16, 17, 30])), min_count=int(rng.choice([1, 1, 2, 3])),
               epochs=int(rng.choice([1, 1, 2])), table_size=int(rng.choice([64, 997, 20011])), seed=int(rng.integers(1, 1 << 30)))
    mode = str(rng.choice(["plain", "plain", "hs", "blocks"]))
    return ids, NV, cfg, mode


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configuration_bit_exact(dge, oracle, seed):
    import torch
    ids, NV, cfg, mode = _case(seed)
    if mode == "hs" and cfg["negative"] == 0 and cfg["dim"] > 256:
        cfg["dim"] = 64
    n_ranks = 3 if mode == "blocks" else 0
    om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                           seed=cfg["seed"], table_size=cfg["table_size"], arith=1, use_hs=(mode == "hs"), part_n=n_ranks)
    c = dge.make_config(cfg["dim"], cfg["window"], NV, negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                        workers=1, seed=cfg["seed"], table_size=cfg["table_size"], use_hs=(mode == "hs"))
    if mode != "blocks" or om.V < n_ranks:
        if mode == "blocks":
            om = oracle.train_sgns(ids, NV, cfg["dim"], cfg["window"], negative=cfg["negative"], min_count=cfg["min_count"], epochs=cfg["epochs"],
                                   seed=cfg["seed"], table_size=cfg["table_size"], arith=1)
        dm = dge.SgnsModel.fit(ids, c, 0)
        pairs = dm.stats()["pairs"]
    else:
        corpus = dge.WalkCorpus.from_host(ids, 0)
        counts = torch.zeros(NV, dtype=torch.int64, device="cuda:0"); corpus.count_tokens(NV, counts)
        ms = [dge.SgnsModel.create(c, counts, 0) for _ in range(n_ranks)]
        for ep in range(cfg["epochs"]):
            simulate_block_schedule(ms, lambda m: m.train(corpus, epoch=ep))
        simulate_gather_syn0(ms)
        dm = ms[0]
        pairs = sum(m.stats()["pairs"] for m in ms)
    syn0, vid = dm.vectors()
    assert np.array_equal(vid, om.vocab_ids), (seed, mode, cfg)
    assert pairs == om.pairs, (seed, mode, cfg)
    assert np.array_equal(bits(syn0), bits(om.syn0)), (seed, mode, cfg)
    assert np.array_equal(bits(dm.syn1neg()), bits(om.syn1neg)), (seed, mode, cfg)
    if mode == "hs" and om.V > 1:
        assert np.array_equal(bits(dm.syn1()), bits(om.syn1)), (seed, mode, cfg)
