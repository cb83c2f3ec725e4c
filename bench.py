#!/usr/bin/env python3
"""bench.py — SGNS training edges/sec on the synthetic time-sliced flow graph (BASELINE.json metric).

One process per GPU.  A "step" is one pass of the hot path over one batch of walks: the walk kernel re-samples the
batch from the alias tables in HBM, the batch is remapped to vocabulary rows, and the SGNS kernel trains every
(centre, context) pair of it with 1 positive + K negative updates.  Everything the step reads is resident in HBM
before the timed region starts.  With N > 1 ranks the block schedule of embedding_amd/distributed.py runs: rows split by
row % N, N episodes per global batch of N x the single-GPU batch, the syn1neg partitions passed round a ring of point-to-point RCCL
transfers (`--multi-gpu allreduce` = the walk-shard + delta-averaging scheme, kept for comparison).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]/[3]: R*T = 1 000 008 layered vertices, ~100 M edges, 24 hourly slices, D=128, K=5
    "cfg3": dict(R=41667, T=24, mean_degree=100, dim=128, negative=5, L=24, walks_per_vertex=10, expect_policy=5,
                 name="synthetic 1M-node / 100M-edge flow graph, 24 timeslices, dim=128, K=5, L=W=24"),
    # the same shape with Zipf-popular destination regions (real trip data is skewed; the headline stays cfg3 as specified): the
    # vocabulary gets a head, auto moves it out of the lock protocol (policy 7) — printed next to cfg3 in profiles/ and README.md
    "cfg3_zipf": dict(R=41667, T=24, mean_degree=100, dim=128, negative=5, L=24, walks_per_vertex=10, dst="zipf", expect_policy=7,
                      name="synthetic 1M-node / 100M-edge flow graph with Zipf-popular destinations, 24 timeslices, dim=128, K=5, L=W=24"),
    # BASELINE.json configs[1]: 100k-node / 5M-edge static graph, D=64, K=5
    "cfg2": dict(R=100000, T=1, mean_degree=50, dim=64, negative=5, L=8, walks_per_vertex=10, expect_policy=8,
                 name="synthetic 100k-node / 5M-edge static flow graph, dim=64, K=5, L=W=8"),
    # BASELINE.json configs[4]: power-law 10M-node / 1B-edge dynamic graph, D=256, K=20 (8 GPUs in the config; runs on 1)
    "cfg5": dict(R=416667, T=24, n_edges=1_000_000_000, dim=256, negative=20, L=24, walks_per_vertex=1, powerlaw=True, expect_policy=7,
                 name="power-law 10M-node / 1B-edge dynamic graph, 24 timeslices, dim=256, K=20, L=W=24"),
    # BASELINE.json configs[0] at the reference's own size: tract level, 801 regions x 8 slices, D=20, 15.6 M walks
    # (J/DeepWalk.java:62-66,89-104); the flow graph is synthetic (the taxi data is not shipped)
    "cfg1": dict(R=801, T=8, mean_degree=200, dim=20, negative=5, L=8, walks_per_vertex=2434,
                 name="reference-sized tract graph: 801 regions x 8 slices, dim=20, K=5, L=W=8, 15.6 M walks"),
    # small smoke-sized workload for quick checks
    "tiny": dict(R=2000, T=8, mean_degree=20, dim=128, negative=5, L=8, walks_per_vertex=10,
                 name="tiny 16k-node graph (debug)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-walks", type=int, default=0,
                    help="walks per step per GPU.  Default, one GPU: epoch/10.  Default, N ranks of the block schedule: the GLOBAL batch is epoch/10 whatever N "
                         "(what embedding_amd.distributed.fit_distributed steps in: the schedule reproduces the one-GPU embedding only in batches of at most a fifth of "
                         "the epoch, profiles/r04_blocks_quality.txt), i.e. epoch/(10 N) walks per GPU; given explicitly, the global batch is N x this")
    ap.add_argument("--weak-batch", action="store_true",
                    help="N ranks: the round-1..4 step, a global batch of N x epoch/10 walks (8 M at N = 8: 0.8 of the epoch in ONE batch — a batch the product does "
                         "not train with, it costs link-prediction AUC 0.959 -> 0.915; kept to compare per-rank throughput at large batches)")
    ap.add_argument("--workers", type=int, default=0, help="SGNS walk workers (0 = fill the device)")
    ap.add_argument("--multi-gpu", choices=["blocks", "allreduce"], default="blocks",
                    help="N>1: 'blocks' = row-partitioned block schedule, exact (default); 'allreduce' = the earlier walk-shard + delta "
                         "averaging scheme, kept for comparison (it under-trains by the factor N: DESIGN.md §7)")
    ap.add_argument("--ring-transport", choices=["auto", "p2p", "allgather"], default="auto",
                    help="N>1, block schedule: how a trained syn1neg partition reaches the next rank (auto: point to point, all-gather if refused)")
    ap.add_argument("--sim-ranks", type=int, default=0,
                    help="single process: run rank 0's share of an N-rank block-schedule step (its N episodes over the N-fold batch, "
                         "partition pack/unpack included, no network) to estimate per-rank throughput at N ranks")
    ap.add_argument("--sim-rank", type=int, default=0, help="with --sim-ranks: which rank's episodes to run (its syn0 partition; default 0, the busiest rows' owner)")
    ap.add_argument("--sim-episode-times", action="store_true",
                    help="with --sim-ranks: after the timed steps, one more step with a host wait per episode; kernel ms and pairs of every block go to stderr (diagnostic)")
    ap.add_argument("--dim", type=int, default=0, help="override the workload's embedding dimension (experiments)")
    ap.add_argument("--negative", type=int, default=-1, help="override the workload's negative count (experiments)")
    ap.add_argument("--hs", action="store_true", help="train the hierarchical-softmax term as well (dge_train_config.use_hs; not the headline path)")
    ap.add_argument("--policy", type=int, default=0, help="dge_train_config.update_policy (0 auto = float atomics)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--force-exchange", action="store_true", help="run the delta exchange even with one rank (plumbing check)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo lets several ranks share one GPU (debugging the multi-rank plumbing)")
    ap.add_argument("--placement-candidates", type=int, default=4,
                    help="the library's placement search before the timed region (dge_model_tune_placement: each large array of the model is tried "
                         "in up to N - 1 other allocations on a quarter-batch probe launch, the faster placement stays; tables restored afterwards; "
                         "which physical memory an array received is worth up to 15 %%: profiles/r03_placement.txt); 1 = whatever the first allocation got")
    ap.add_argument("--placement-trials", type=int, default=1,
                    help="round 2's whole-model form of the same (SgnsModel.create_placed: this many models side by side, the fastest kept); default 1 = off")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="dge_set_tuning knob for experiments (hot_rows, hs_drain, sorted_chunk, sorted_walks, ...)")
    ap.add_argument("--epoch", action="store_true",
                    help="one GPU: DeepWalk.main end to end (J/DeepWalk.java:120-140) instead of steady-state steps — edge tuples in HBM -> CSR -> alias tables "
                         "-> the epoch's walks -> the one-shot fit (counts, vocabulary, unigram table, tables incl. their placement probes, the placement search "
                         "where it pays, ONE epoch with the real decaying learning rate) -> the .vec text file; wall clock per stage; the CPU restatement's "
                         "projected end-to-end time beside it.  Prints one JSON line (not the driver's metric line).")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="ranks only meet (init_process_group, all-reduce, barrier) and rank 0 prints {\"n_gpus\": N, ...}: checks the "
                         "launcher on a box without a GPU; nothing is measured")
    args = ap.parse_args()

    # `python bench.py --gpus N` (the driver's command shape) with no torchrun environment: start the N ranks here.  The
    # parent touches neither torch nor HIP, it waits for the children and exits with their status; rank 0 of the children
    # prints the JSON line on the inherited stdout.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or run `python bench.py --gpus N`, "
                 "which starts the ranks itself)" % (args.gpus, world_env))
    if args.rendezvous_check:
        return rendezvous_check(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = world_env
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # several ranks on one card (debug only)
    # a rank without a device of its own must say so and leave before the rendezvous: the other ranks would wait for it in init_process_group
    n_dev = torch.cuda.device_count()
    if n_dev <= local_rank:
        sys.exit("bench.py: rank %d needs GPU %d but this node shows %d device(s) (--gpus %d): nothing measured" % (rank, local_rank, n_dev, N))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    n_ranks_seen = 1
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=N)
        # `n_gpus` of the JSON line is what the collective backend itself counted, not the flag
        one = torch.ones(1, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        n_ranks_seen = int(one.item())
        if n_ranks_seen != N:
            sys.exit("bench.py: %d ranks answered the all-reduce, --gpus %d" % (n_ranks_seen, N))

    import embedding_amd as E
    from embedding_amd import synth
    from embedding_amd.distributed import RingTransport, allreduce_counts, block_schedule_step, exchange_deltas, shard_plan

    for kv in args.tune:
        k, v = kv.split("=")
        E._native.check(E.lib.dge_set_tuning(E.engine.TUNING_KNOBS[k], int(v)))

    def stage(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.time()
    wl = dict(WORKLOADS[args.workload])
    if args.scale != 1.0:          # shrink a workload (debugging): fewer regions and edges, same shape
        wl["R"] = max(16, int(wl["R"] * args.scale))
        if "n_edges" in wl:
            wl["n_edges"] = int(wl["n_edges"] * args.scale)
    if args.dim > 0:
        wl["dim"] = args.dim; wl["name"] += " [dim=%d]" % args.dim
    if args.negative >= 0:
        wl["negative"] = args.negative; wl["name"] += " [K=%d]" % args.negative
    R, T, L, D, K = wl["R"], wl["T"], wl["L"], wl["dim"], wl["negative"]
    NV = R * T
    t0 = time.time()

    if args.epoch:
        if N != 1:
            sys.exit("bench.py --epoch is a one-GPU measurement")
        return run_epoch(args, wl, E, synth, local_rank, dev, stage)

    # ---- setup (untimed): replicate the edge store on every GPU, build alias tables
    if wl.get("powerlaw"):
        G = synth.powerlaw_flow_graph_torch(R, T, wl["n_edges"], dev)
    else:
        G = synth.flow_graph_torch(R, T, wl["mean_degree"], dev, dst=wl.get("dst", "uniform"))
    stage("edge list generated: %d edges" % G["n_edges"])
    g = E.DeviceGraph(local_rank)
    g.add_edges_device(G["src"], G["dst"], G["w"])
    n_edges = G["n_edges"]
    sources = G["sources"] if T > 1 else np.arange(R, dtype=np.int32)
    del G
    torch.cuda.empty_cache()
    stage("edges copied into the store")
    g.set_sources(sources)
    stage("CSR built, sources set")
    g.build_alias(exact=False)       # Vose pairing: same distribution as the reference order, O(k) for hubs
    stage("alias tables built")

    # ---- setup: this rank's shard of the epoch corpus, global vocabulary
    epoch_walks = wl["walks_per_vertex"] * NV
    shard0, shard = shard_plan(epoch_walks, N, rank)
    WALK_SEED = 20171106
    corpus = g.sample_walks_device(shard, L, seed=WALK_SEED, rng_mode=1, first_index=shard0)
    stage("epoch corpus sampled: %d walks" % shard)
    counts = torch.zeros(NV, dtype=torch.int64, device=dev)
    corpus.count_tokens(NV, counts)
    allreduce_counts(counts)
    # epochs only sets the learning-rate horizon (alpha decays over epochs*total_words); the bench steps stay near alpha0
    cfg = E.make_config(D, L, NV, negative=K, min_count=2, epochs=1000, workers=args.workers, seed=1, update_policy=args.policy, use_hs=args.hs)
    model = E.SgnsModel.create(cfg, counts, local_rank)
    stage("vocabulary, unigram table and weights ready")
    total_words = int(counts.sum().item())
    NB = args.sim_ranks if (N == 1 and args.sim_ranks > 1) else N          # ranks of the block schedule
    blocks = NB > 1 and args.multi_gpu == "blocks"
    B = args.batch_walks or max(1, epoch_walks // 10)
    if blocks and not args.batch_walks and not args.weak_batch:
        B = max(1, epoch_walks // 10 // NB)                                # the global batch stays epoch/10 (fit_distributed's step): strong scaling
    B = min(B, shard if not blocks else epoch_walks)
    delta = None
    if blocks:
        # the global batch is NB x the per-rank batch; every rank samples ALL of it from its replica of the
        # graph (walk i is the same walk on any rank) and trains its row blocks of it
        del corpus
        torch.cuda.empty_cache()
        BG = B * NB
        corpus = g.sample_walks_device(BG, L, seed=WALK_SEED, rng_mode=1, first_index=0)
        pf = model.partition_floats(NB)
        part_buf = torch.empty(pf, dtype=torch.float32, device=dev)
        recv_buf = torch.empty(pf, dtype=torch.float32, device=dev)
        stage("block schedule: %d ranks, global batch %d walks, partition buffers %.0f MB" % (NB, BG, pf * 4 / 1e6))
    # start-up placement trials (SgnsModel.create_placed): the same launch the steps will run, on models created side by side; the fastest stays
    trial_ms = None
    if args.placement_trials > 1:
        def probe(m):
            if blocks:
                m.set_partition(NB, rank if N > 1 else 0, rank if N > 1 else 0)
            ms = float("inf")
            for _ in range(2):                                 # (the first launch of a model also allocates its work buffers)
                m.reset_stats()
                m.train(corpus, 0, BG if blocks else B, walk_index_base=0, epoch=0, words_before=0, words_scale=1.0, total_walks=epoch_walks)
                ms = min(ms, m.stats()["kernel_ms"])
            if blocks:
                m.set_partition(1)
            m.reset_stats()
            return ms
        model, trial_ms = E.SgnsModel.create_placed(cfg, counts, local_rank, probe, trials=args.placement_trials, first=model)
        stage("placement trials: %s ms per launch, kept the fastest" % ", ".join("%.1f" % x for x in trial_ms))
    placement = None
    if args.placement_candidates > 1:
        n_probe = max(min(BG if blocks else B, 1024), (BG if blocks else B) // 4)
        if blocks:
            model.set_partition(NB, rank if N > 1 else 0, rank if N > 1 else 0)
        before, after, moved = model.tune_placement(corpus, 0, n_probe, candidates=args.placement_candidates)
        if blocks:
            model.set_partition(1)
        placement = {"candidates": args.placement_candidates, "probe_walks": n_probe, "probe_ms_before": round(before, 2), "probe_ms_after": round(after, 2),
                     "arrays_moved": moved}
        stage("placement search: probe launch %.1f -> %.1f ms, %d arrays moved" % (before, after, moved))
    ring = None
    if blocks and N > 1:           # how partitions travel: decided once, by all ranks together (a probe transfer, verdict all-reduced)
        ring = RingTransport.choose(N, rank, dist, device=dev, requested=None if args.ring_transport == "auto" else args.ring_transport)
        stage("ring transport: %s" % ring.mode)
    exchange = (N > 1 and not blocks) or args.force_exchange
    if exchange and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": torch.device(dev)} if args.backend == "nccl" else {}))
    if exchange:
        delta = torch.empty(model.sync_size(), dtype=torch.float32, device=dev)
        model.snapshot()
    copy_rate = box_copy_rate(dev) if rank == 0 else None
    row_rates = None
    if rank == 0 and NV * (-(-D // 64) * 64) * 4 < 0xFFFFFFFF:      # (the probe addresses a table through one descriptor)
        row_rates = model.row_rates()        # this model's own tables: random rows read / read + written back (GB/s), lock exchanges, table look-ups per s
    setup_s = time.time() - t0

    episode_log = None

    def step_blocks(i):
        first = (i * BG) % max(epoch_walks - BG + 1, 1)        # global index of the batch's first walk
        state = {"sampled": False}

        def train_fn():
            if not state["sampled"]:                           # episode 0 also samples the batch
                state["sampled"] = True
                model.walk_and_train(g, corpus, 0, BG, walk_seed=WALK_SEED, walk_index_base=first, epoch=0, words_before=0,
                                     words_scale=1.0, total_walks=epoch_walks)
            else:
                model.train(corpus, 0, BG, walk_index_base=first, epoch=0, words_before=0, words_scale=1.0, total_walks=epoch_walks)

        if N > 1:
            block_schedule_step(model, train_fn, N, rank, part_buf, recv_buf, transport=ring)
        else:                                                  # --sim-ranks: rank 0's episodes, exchange replaced by a local pack/unpack
            r0 = args.sim_rank % NB
            for e in range(NB):
                t_part = (r0 + e) % NB
                model.set_partition(NB, r0, t_part)
                if episode_log is not None:
                    before = model.stats()
                train_fn()
                if episode_log is not None:
                    after = model.stats()
                    episode_log.append((r0, t_part, after["kernel_ms"] - before["kernel_ms"], after["pairs"] - before["pairs"]))
                ms = model.stream()                          # (pack and unpack stay on the library's stream, as a hand-off's would; no host wait)
                model.export_partition_async(1, NB, t_part, part_buf, ms)
                model.import_partition_async(1, NB, t_part, part_buf, ms)
            model.set_partition(1)

    def step(i):
        if blocks:
            return step_blocks(i)
        row0 = (i * B) % max(shard - B + 1, 1)
        model.walk_and_train(g, corpus, row0, B, walk_seed=WALK_SEED, walk_index_base=shard0 + row0, epoch=0,
                             words_before=0, words_scale=float(N), total_walks=epoch_walks)
        if exchange:
            exchange_deltas(model, delta, N)

    def sync():
        model.stats()                 # drains the library's stream
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    sync()
    stage("warm-up done")
    model.reset_stats()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    dt = time.perf_counter() - t1
    st = model.stats()
    if blocks and N == 1 and args.sim_episode_times:
        episode_log = []
        step(args.warmup + args.steps)
        for r0, t_part, ms_e, pairs_e in episode_log:
            print("[bench] block (syn0 partition %d, syn1neg partition %d): %8.2f ms  %10d pairs  %.3e pairs/s" % (r0, t_part, ms_e, pairs_e, pairs_e / max(ms_e, 1e-9) * 1e3),
                  file=sys.stderr, flush=True)
        episode_log = None

    pairs = torch.tensor([float(st["pairs"])], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if N > 1:
        dist.all_reduce(pairs)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_pairs = float(pairs.item())
    elapsed = float(tmax.item())

    if rank == 0:
        value = total_pairs / elapsed
        bytes_per_pair = 8 * D * (K + 2)                      # one syn0 row + K+1 syn1neg rows, read+written
        hs_path = None
        if args.hs:
            # + the inner-node rows on the centre's Huffman path (count-weighted mean length), ONCE PER CENTRE: the path belongs to the centre, all its contexts
            # meet the same nodes, so what the term must move per pair is path rows / contexts per centre (round 5: until then the rows were charged per pair —
            # what the pair-by-pair kernel moves, not what the algorithm needs — and the wave-per-centre kernel printed a fraction above 1)
            off, _, _ = model.huffman()
            cnt = model.counts().astype(np.float64)
            hs_path = float((np.diff(off) * cnt).sum() / max(cnt.sum(), 1.0))
            ctx_per_centre = st["pairs"] / max(st["words"], 1)
            bytes_per_pair += 8 * D * hs_path / max(ctx_per_centre, 1.0)
        launches = max(st["launches"], 1)
        ms_per_launch = st["kernel_ms"] / launches
        pairs_per_launch = st["pairs"] / launches
        achieved = pairs_per_launch * bytes_per_pair / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        sched = model.schedule()
        stamp = E.engine.build_stamp()
        kernel_name = model.kernel()                         # what the library's latest launch ran (include/dge.h: dge_model_kernel)
        out = {
            "metric": "SGNS training edges/sec",
            "value": value,
            "unit": "edges/s",
            "n_gpus": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # one GPU, and --weak-batch: per-GPU work fixed; N ranks at the default: the GLOBAL batch is fixed at epoch/10 (the batch size that keeps the embedding)
            "scaling": "strong" if (blocks and not args.batch_walks and not args.weak_batch) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "vertices": NV, "edges": int(n_edges), "timeslices": T, "dim": D,
                       "negatives": K, "walk_len": L, "window": L, "walks_per_step_per_gpu": B,
                       "global_batch_walks": BG if blocks else B * N, "batch_fraction_of_epoch": round((BG if blocks else B * N) / epoch_walks, 4),
                       "pairs_per_step_per_gpu": st["pairs"] / args.steps, "vocabulary": int((counts >= 2).sum().item()),
                       "sgns_workers": args.workers, "table_placement": model.table_placement(), "table_runs": model.table_runs(), "placement_search": placement, "placement_trial_ms": trial_ms and [round(x, 1) for x in trial_ms],
                       "lr_horizon_epochs": 1000, "ring_transport": ring and ring.mode, "update_policy": args.policy, "use_hs": bool(args.hs), "parallelism": ("block schedule x%d: rows split by row %% N, N episodes per global batch, syn1neg partitions passed round a ring" % N if blocks and N > 1
                                       else "SIMULATED rank 0 of a %d-rank block schedule on one GPU (value = this rank's share only)" % NB if blocks
                                       else "walk-shard x%d, RCCL all-reduce of deltas per step (comparison mode)" % N if N > 1 else "1 GPU"),
                       "lock_stats": model.lock_stats() if blocks else None,      # one block under the lock kernels: pairs put back / rounds that left rows unwon / rounds
                       "setup_s": round(setup_s, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None if (args.dim or args.negative >= 0 or NB > 1) else measured_traffic(args.workload, ("hs_centre" if "hsw" in kernel_name else "hs_pairs") if args.hs else "policy%d" % sched["update_policy"], pairs_per_launch, stamp),
                         "traffic_source": "profiles/traffic.json: bytes per pair from the committed rocprofv3 PMC passes of this workload x the pairs of this run (not counters of this run); null unless those passes were collected with the kernels of the loaded library (build stamp %s)" % json.dumps(stamp),
                         "kernel": kernel_name, "schedule": sched,
                         "ms_per_launch": ms_per_launch,
                         "bytes_per_pair": bytes_per_pair, "pairs_per_launch": pairs_per_launch,
                         "hs_path_rows_per_centre": hs_path,
                         "walk_kernel_ms_per_launch": st["walk_kernel_ms"] / launches,
                         "walk_steps_per_s": (B * L) / (st["walk_kernel_ms"] / launches * 1e-3) if st["walk_kernel_ms"] > 0 else None,
                         "walk_bytes_per_step": 36, "box_copy_GBps": copy_rate,
                         "row_read_GBps": row_rates and round(row_rates[0], 1), "row_rewrite_GBps": row_rates and round(row_rates[1], 1),
                         # what the lock kernel actually runs against (profiles/r03_shape_sweep.txt): requests at the L2's memory side — reads leave as 128 bytes,
                         # writes as 64 — at ~8.5e10/s in every shape measured; from the committed PMC passes like `traffic`, not from counters of this run
                         "fabric_requests_per_s": None if (args.dim or args.negative >= 0 or NB > 1) else measured_requests(args.workload, ("hs_centre" if "hsw" in kernel_name else "hs_pairs") if args.hs else "policy%d" % sched["update_policy"], pairs_per_launch, ms_per_launch, stamp)},
        }
        if "expect_policy" in wl and not (args.policy or args.workers or args.hs or NB > 1) and sched["update_policy"] != wl["expect_policy"]:
            print("warning: workload %s resolved to policy %d, the committed traffic profile is for policy %d" % (args.workload, sched["update_policy"], wl["expect_policy"]), file=sys.stderr)
        if N == 1 and not args.no_cpu_baseline:
            sample = g.sample_walks(min(shard, 400_000), L, seed=WALK_SEED, rng_mode=1, first_index=shard0)   # = first rows of the corpus
            out["cpu_baseline"] = cpu_baseline(sample, NV, D, L, K, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def run_epoch(args, wl, E, synth, local_rank, dev, stage):
    """DeepWalk.main (J/DeepWalk.java:120-140) end to end on one GPU, wall clock: checkInputFile's walk generation
    (J/CrossTimeGraph.java:115-124: construct the graph, initiateAliasTables, numSamples walks) and learnEmbedding (:32-83: vocabulary, fit(), writeWordVectors).
    The input — the (src, dst, weight) tuples the reference reads out of its flow maps — is resident in HBM when the clock starts."""
    import tempfile

    import numpy as np
    import torch
    R, T, L, D, K = wl["R"], wl["T"], wl["L"], wl["dim"], wl["negative"]
    NV = R * T
    G = synth.powerlaw_flow_graph_torch(R, T, wl["n_edges"], dev) if wl.get("powerlaw") else synth.flow_graph_torch(R, T, wl["mean_degree"], dev, dst=wl.get("dst", "uniform"))
    n_edges = int(G["n_edges"])
    sources = G["sources"] if T > 1 else np.arange(R, dtype=np.int32)
    epoch_walks = wl["walks_per_vertex"] * NV
    torch.cuda.synchronize()
    stage("edge tuples resident: %d" % n_edges)
    st = {}

    def timed(name, fn):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); st[name] = time.perf_counter() - t
        stage("%s: %.3f s" % (name, st[name]))
        return r

    g = E.DeviceGraph(local_rank)
    timed("store_s", lambda: (g.add_edges_device(G["src"], G["dst"], G["w"]), g.set_sources(sources)))
    del G
    timed("alias_s", lambda: g.build_alias(exact=False))
    corpus = timed("walks_s", lambda: g.sample_walks_device(epoch_walks, L, seed=20171106, rng_mode=1, first_index=0))
    cfg = E.make_config(D, L, NV, negative=K, min_count=2, epochs=1, workers=args.workers, seed=1, update_policy=args.policy, use_hs=args.hs)
    model = timed("fit_s", lambda: E.SgnsModel.fit(corpus, cfg, local_rank))
    stats, sched, search = model.stats(), model.schedule(), model.placement_search()
    vec = os.path.join(tempfile.gettempdir(), "bench_epoch_%d.vec" % os.getpid())
    timed("vec_s", lambda: model.write_vec(vec))
    vec_bytes = os.path.getsize(vec)
    os.unlink(vec)
    total = sum(st.values())
    out = {"metric": "DeepWalk.main end to end, one epoch (edges trained per wall-clock second)", "value": stats["pairs"] / total, "unit": "edges/s", "n_gpus": 1,
           "higher_is_better": True, "dtype": "f32", "data": "synthetic", "epoch_s": total, "stages_s": {k: round(v, 3) for k, v in st.items()},
           "train_kernel_s": stats["kernel_ms"] * 1e-3, "pairs": stats["pairs"], "walks": epoch_walks, "vec_bytes": vec_bytes,
           "steady_state_edges_per_s": stats["pairs"] / (stats["kernel_ms"] * 1e-3) if stats["kernel_ms"] else None,
           "config": {"workload": wl["name"], "vertices": NV, "edges": n_edges, "dim": D, "negatives": K, "walk_len": L, "epochs": 1, "schedule": sched,
                      "placement_search": search, "table_placement": model.table_placement(), "use_hs": bool(args.hs)}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_epoch_baseline(wl, n_edges, epoch_walks, stats["pairs"], g, args.cpu_seconds)
    print(json.dumps(out), flush=True)


def cpu_epoch_baseline(wl, n_edges, epoch_walks, epoch_pairs, g, seconds):
    """The CPU restatement end to end, PROJECTED from bounded samples (the full epoch would take the oracle ~20 minutes): edge store + alias tables and the walk
    sampler on the same kind of graph at 1/16 of the regions (both linear in edges / walk steps; single thread, as the reference's walk half is), training
    on a sample of the device's walks with every core of one GPU's host share (Hogwild; the reference uses 8 workers).  kind "port": not the Java path."""
    import numpy as np
    from embedding_amd import synth
    from oracle import oracle as O
    O.build()
    Rs = max(64, wl["R"] // 16)
    Gs = synth.flow_graph_numpy(Rs, wl["T"], wl.get("mean_degree", 100))
    t = time.perf_counter()
    og = O.Graph(); og.add_edges(Gs["src"], Gs["dst"], Gs["w"]); og.set_sources(Gs["sources"] if wl["T"] > 1 else np.arange(Rs, dtype=np.int32)); og.build_alias(False)
    build_rate = len(Gs["src"]) / (time.perf_counter() - t)                       # edges per second into store + tables
    nw = 200_000
    t = time.perf_counter(); og.sample_walks(nw, wl["L"], seed=1, rng_mode=0); walk_rate = nw / (time.perf_counter() - t)
    del og
    sample = g.sample_walks(min(epoch_walks, 400_000), wl["L"], seed=20171106, rng_mode=1, first_index=0)
    tr = cpu_baseline(sample, wl["R"] * wl["T"], wl["dim"], wl["L"], wl["negative"], seconds)
    proj = n_edges / build_rate + epoch_walks / walk_rate + epoch_pairs / tr["value"]
    return {"value": epoch_pairs / proj, "unit": "edges/s", "cores": tr["cores"], "kind": "port", "projected_epoch_s": proj,
            "parts": {"store_and_alias_edges_per_s": build_rate, "walks_per_s_one_thread": walk_rate, "train_edges_per_s": tr["value"]},
            "sample": "store + alias tables + %d walks on a %d-region x %d-slice graph of the same generator (one thread), training: %s; projected linearly to the epoch; .vec output not included"
                      % (nw, Rs, wl["T"], tr["sample"])}


def launch_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run on this node (one per GPU, rendezvous on 127.0.0.1 at a free
    port) and return their exit status.  Called before anything imports torch: the parent process never initialises a GPU."""
    import socket
    import subprocess
    # under rocprofv3 this process already carries the profiler's preloaded library (and with it an initialised GPU): starting ranks from
    # here would be the launcher hop this pool forbids.  Profile one rank per rocprofv3 process instead (scripts/collect_profiles.sh).
    if any("rocprofiler" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or os.environ.get("ROCPROF_OUTPUT_PATH"):
        print("bench.py: --gpus %d under a profiler: run one rank per rocprofv3 process (export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)" % n, file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rendezvous_check(args):
    """--rendezvous-check: the ranks meet over the chosen backend and count themselves; no GPU work, nothing measured."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = args.backend if (args.backend == "gloo" or torch.cuda.is_available()) else "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    one = torch.ones(1, dtype=torch.int64, device="cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(one)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "SGNS training edges/sec", "value": None, "unit": "edges/s", "n_gpus": int(one.item()),
                          "rendezvous": "ok", "backend": backend}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def _traffic_entry(workload, policy, stamp):
    """The committed counter profile of (workload, schedule) — or None when there is none, or when it was collected with OTHER kernels than the
    loaded library's: every entry of profiles/traffic.json carries the source hash of the build it was measured on (`stamp`, written by
    scripts/traffic_update.py; include/dge.h: dge_build_stamp), `stamp` here is {"kernels": ..., "sorted": ...} of the loaded library.
    stamp=None skips the comparison (tests)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(path)).get("%s/%s" % (workload, policy))
    except (OSError, ValueError):
        return None
    if not e or "bytes_per_pair" not in e:
        return None
    if stamp is not None and e.get("stamp") != stamp.get("sorted" if policy == "policy8" else "kernels"):
        return None
    return e


def measured_traffic(workload, policy, pairs_per_launch, stamp=None):       # policy: "policy5", "policy7", "hs", ...
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE / WRITE_SIZE per pair of the same workload and policy, corrected as profiles/README.md describes).
    bench.py cannot collect counters itself; None when no profile of this configuration AND of this build's kernels is committed."""
    e = _traffic_entry(workload, policy, stamp)
    return None if e is None else e["bytes_per_pair"] * pairs_per_launch


def measured_requests(workload, policy, pairs_per_launch, ms_per_launch, stamp=None):
    """Requests per second at the L2's memory side: requests per pair from the committed PMC passes (profiles/traffic.json) x this run's pair rate."""
    e = _traffic_entry(workload, policy, stamp)
    return None if not e or "requests_per_pair" not in e or not ms_per_launch else e["requests_per_pair"] * pairs_per_launch / (ms_per_launch * 1e-3)


def cpu_baseline(walks, NV, D, L, K, seconds):
    """The CPU restatement (oracle, 'port') timed with Hogwild threads on this box's host cores, on a bounded sample
    of the same walks: once with the reference's own `.workers(8)` (J/DeepWalk.java:75) and once with every core of one GPU's
    share of the host.  A reported baseline, not the target; not the Java reference (no JDK here)."""
    from oracle import oracle as O
    O.build()
    # the GPU box exposes every host core but one GPU's share is 16 of them (task notes); more threads than that only
    # time-slice against the cgroup quota (measured: 256 threads run 1.3e6 edges/s, slower than 16)
    cores = min(len(os.sched_getaffinity(0)), 16)

    def run(threads, secs):
        probe = walks[:20_000]
        m = O.train_sgns(probe, NV, D, L, negative=K, threads=threads, table_size=10_000_000)
        rate = m.pairs / max(m.seconds, 1e-9)
        n = int(min(len(walks), max(20_000, 0.5 * secs * rate / (m.pairs / len(probe)))))   # the probe over-estimates the rate ~2x
        m = O.train_sgns(walks[:n], NV, D, L, negative=K, threads=threads, table_size=10_000_000)
        return {"value": m.pairs / max(m.seconds, 1e-9), "unit": "edges/s", "cores": threads, "kind": "port",
                "sample": "%d walks (%d pairs) of the same corpus, oracle/dge_oracle.c Hogwild with %d OpenMP threads, %.1f s"
                          % (n, m.pairs, threads, m.seconds)}

    out = run(cores, 0.6 * seconds)
    if cores != 8:
        w8 = run(min(8, cores), 0.4 * seconds)
        out["reference_workers_8"] = {k: w8[k] for k in ("value", "cores", "sample")}     # .workers(8), J/DeepWalk.java:75
    return out


def box_copy_rate(dev):
    """How fast THIS box moves bytes: a 1 GiB device-to-device copy, read + written bytes per second (GB/s).  Boxes of the pool differ
    by +-7 %; the figure lets a reader tell a slow kernel from a slow box.  Outside the timed region."""
    import torch
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    a.fill_(1.0); b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(5):
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return best


if __name__ == "__main__":
    main()
