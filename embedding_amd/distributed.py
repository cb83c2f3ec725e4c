"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Walk SAMPLING shards by walk index with no collective: the strided RNG makes walk i the same walk on any rank, so any
rank can produce any range of an epoch's walks from its replica of the graph.  Token counts are summed once so that
every rank builds the same vocabulary.

TRAINING uses the block schedule (`block_schedule_step`): vocabulary rows are split by row % N; in episode e rank g
trains, over the SAME global batch of walks, the pairs whose context row is in partition g and whose centre row is in
partition (g+e) % N, with negatives moved into that partition.  The N blocks of an episode are row-disjoint in both
tables and after N episodes every pair was trained exactly once: the result is the single-GPU result with the pairs in
another order — nothing is averaged.  Between episodes every rank passes the syn1neg partition it just trained (V/N rows) to
rank-1, which trains it next (a ring of point-to-point transfers); syn0 partitions stay home until `gather_table`.

The earlier scheme — every rank trains its own walk shard from a snapshot, deltas are all-reduced and applied with 1/N
(`exchange_deltas`) — is kept for comparison only: measured (scripts/quality_exchange.py, profiles/r01_quality_exchange.txt)
it under-trains by the factor N (each row moves 1/N as far per epoch; AUC 0.97 -> 0.91 at N=4, 0.48 at N=8), and summing
instead of averaging overshoots (AUC 0.53 at N=4).  The reference is single-host; this layer is new.
"""


HOST_WAITS = {"n": 0}      # blocking device waits made by this module (tests: an episode of the block schedule makes none)


def _wait_device(t):
    """torch.distributed enqueues collectives on torch's streams; libdge.so works on its own stream.  Outside the block schedule's episodes (counts, delta
    exchange, the final gather) the host simply waits for the tensor's device before handing the pointer over."""
    if getattr(t, "is_cuda", False):
        import torch
        HOST_WAITS["n"] += 1
        torch.cuda.synchronize(t.device)


def _stream_of(t):
    """torch's current stream on the tensor's device as a hipStream_t integer (what libdge's stream-ordered entry points take); None for a host tensor."""
    if getattr(t, "is_cuda", False):
        import torch
        return int(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def shard_plan(epoch_walks, world, rank):
    """(first_walk_index, n_walks) of `rank`: contiguous, disjoint, covering [0, world*(epoch_walks//world))."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    n = epoch_walks // world
    return rank * n, n


def allreduce_counts(counts, dist_mod=None):
    """Sum per-rank token counts in place (device int64 tensor) -> identical vocabulary on every rank."""
    import torch.distributed as dist
    d = dist_mod or dist
    if d.is_initialized() and d.get_world_size() > 1:
        d.all_reduce(counts)
        _wait_device(counts)
    return counts


def exchange_deltas(model, buf, world, dist_mod=None):
    """delta = current - snapshot on every rank; all-reduce(sum); current = snapshot + delta_sum / world.

    `model` needs export_delta(buf) / import_delta(buf, scale) (embedding_amd.SgnsModel); `buf` is a float32 tensor of
    model.sync_size() elements on the model's device."""
    import torch.distributed as dist
    d = dist_mod or dist
    model.export_delta(buf)            # returns after the library's stream has drained
    if world > 1 or (d.is_initialized() and getattr(buf, "is_cuda", False)):
        d.all_reduce(buf)
        _wait_device(buf)              # the collective runs on torch's stream; libdge reads buf on its own stream
    model.import_delta(buf, 1.0 / world)


class RingTransport:
    """How a trained syn1neg partition reaches the next rank — decided ONCE, by all ranks together, never inside a step.

    mode "p2p": one point-to-point transfer per rank and episode; "allgather": the same bytes through an all-gather.
    `choose` runs a four-float point-to-point probe on every rank, time-boxed, and all-reduces (MIN) the outcome over a control group
    (a gloo group when the data backend is nccl: a refused or hung RCCL transfer must not be what carries the verdict): all ranks
    succeeded -> "p2p"; any rank was refused (an exception) -> every rank takes "allgather"; any rank's probe did not finish within the
    time box -> RuntimeError on every rank (the communicator is not to be trusted: exit non-zero rather than hang).  After the
    decision a failing transfer raises; no rank ever switches mode on its own."""

    def __init__(self, mode, ctrl=None):
        if mode not in ("p2p", "allgather"):
            raise ValueError("ring transport %r" % (mode,))
        self.mode = mode
        self.ctrl = ctrl

    @classmethod
    def choose(cls, world, rank, d, device="cpu", requested=None, timeout_s=60.0):
        import torch
        if requested in ("p2p", "allgather"):
            return cls(requested)
        ctrl = None
        if d.get_backend() == "nccl":
            ctrl = d.new_group(backend="gloo")             # (collective: every rank creates it)
        d.barrier(group=ctrl)                              # the time box below starts on every rank together (ranks arrive from set-up work of different length)
        ok, why = 1, ""
        try:
            staged = str(device).startswith("cuda") and d.get_backend() != "nccl"
            s = torch.full((4,), float(rank), dtype=torch.float32, device="cpu" if staged else device)
            r = torch.empty(4, dtype=torch.float32, device="cpu" if staged else device)
            works = d.batch_isend_irecv([d.P2POp(d.isend, s, (rank - 1) % world), d.P2POp(d.irecv, r, (rank + 1) % world)])
            # time box: the waits run on a helper thread (gloo only completes a transfer inside wait(); an RCCL transfer whose peer never
            # posts its half would block a stream synchronisation for good)
            import threading
            done = {}

            def _wait_all():
                try:
                    for w in works:
                        w.wait()
                    if str(device).startswith("cuda"):
                        torch.cuda.synchronize(device)
                    done["ok"] = True
                except Exception as e:  # noqa: BLE001 (reported through the verdict)
                    done["err"] = e
            th = threading.Thread(target=_wait_all, daemon=True)
            th.start(); th.join(timeout_s)
            if th.is_alive():
                ok, why = -1, "no completion within %.0f s" % timeout_s
            elif "err" in done:
                raise RuntimeError(str(done["err"]))
            if ok == 1:
                if float(r[0]) != float((rank + 1) % world):
                    ok, why = 0, "probe payload wrong"
        except (RuntimeError, NotImplementedError, ValueError) as e:
            ok, why = 0, (str(e).splitlines() or ["?"])[0]
        flag = torch.tensor([ok], dtype=torch.int32)
        d.all_reduce(flag, op=d.ReduceOp.MIN, group=ctrl)          # (a CPU tensor: the gloo control group under nccl, the default group otherwise)
        verdict = int(flag.item())
        if verdict < 0:
            raise RuntimeError("ring transport: the point-to-point probe hung on at least one rank (rank %d: %s) — not falling back on a "
                               "communicator in that state" % (rank, why or "ok here"))
        if verdict == 0:
            import sys
            if why or rank == 0:
                print("[distributed] rank %d: point-to-point transfers unavailable%s: every rank uses all-gather" % (rank, " (%s)" % why if why else ""),
                      file=sys.stderr, flush=True)
            return cls("allgather", ctrl)
        return cls("p2p", ctrl)


def _ring_pass(send_buf, recv_buf, world, rank, d, transport):
    """Every rank sends `send_buf` to rank - 1 and receives rank + 1's into `recv_buf` (one point-to-point transfer per rank: over
    xGMI a partition of cfg3, 64 MB, is ~0.5 ms on one link).  Backends that cannot move device tensors point to point (gloo in the
    CPU/one-GPU tests) are staged through host memory.  transport.mode "allgather" moves the same bytes with an all-gather and keeps
    rank + 1's share.  The mode was fixed for the whole run by RingTransport.choose; a failure here raises."""
    import torch
    dst, src = (rank - 1) % world, (rank + 1) % world
    if transport.mode == "p2p":
        staged = getattr(send_buf, "is_cuda", False) and d.get_backend() != "nccl"
        s, r = (send_buf.cpu(), torch.empty(recv_buf.shape, dtype=recv_buf.dtype)) if staged else (send_buf, recv_buf)
        for w in d.batch_isend_irecv([d.P2POp(d.isend, s, dst), d.P2POp(d.irecv, r, src)]):
            w.wait()
        if staged:
            recv_buf.copy_(r)
    else:
        allb = torch.empty(send_buf.numel() * world, dtype=send_buf.dtype, device=send_buf.device)
        d.all_gather_into_tensor(allb, send_buf)
        recv_buf.copy_(allb[src * send_buf.numel():(src + 1) * send_buf.numel()])
    # (no host wait: under nccl `wait()` orders torch's current stream behind the transfer, and the import that follows makes libdge's stream wait for that
    #  stream — embedding_amd.SgnsModel.import_partition_async; the staged gloo path has completed on the host already)


def block_schedule_step(model, train_fn, world, rank, part_buf=None, recv_buf=None, dist_mod=None, transport=None):
    """One global batch under the block schedule.  `train_fn()` trains the batch on `model` (it is called once per
    episode, with the model's partition filter set); `model` needs set_partition / export_partition / import_partition /
    partition_floats (embedding_amd.SgnsModel).  `part_buf` and `recv_buf` are float32 tensors of partition_floats elements on the
    model's device (allocated here when not given).  Returns the buffers for reuse.

    The syn1neg partitions travel a RING: partition p is trained by rank p in episode 0, by rank p-1 in episode 1, ... so after
    every episode a rank hands the partition it just trained to rank-1 and takes the next one from rank+1 — one partition per
    rank and episode, point to point, instead of an all-gather of all of them; a rank only ever reads the syn1neg partition it
    is about to train and its own syn0 partition.  After the N-th episode partition p is back on rank p: the invariant the
    next batch starts from (and `gather_table` collects from).

    `transport`: a RingTransport (chosen once per run: RingTransport.choose, a collective), or "p2p" / "allgather"; None chooses here, on
    every rank alike, which costs a probe per call — callers that step repeatedly choose once and pass the object."""
    if world <= 1:
        model.set_partition(1)
        train_fn()
        return part_buf, recv_buf
    import torch
    import torch.distributed as dist
    d = dist_mod or dist
    pf = model.partition_floats(world)
    if part_buf is None or recv_buf is None or recv_buf.numel() != pf:
        dev = getattr(model, "torch_device", None) or "cpu"
        part_buf = torch.empty(pf, dtype=torch.float32, device=dev)
        recv_buf = torch.empty(pf, dtype=torch.float32, device=dev)
    if not isinstance(transport, RingTransport):
        transport = RingTransport.choose(world, rank, d, device=part_buf.device, requested=transport)
    tables = (1, 2) if getattr(getattr(model, "cfg", None), "use_hs", 0) else (1,)     # with the hierarchical softmax the syn1 partition of the same number travels along
    # The hand-off is STREAM-ORDERED (round 5): the pack kernel runs on libdge's stream and torch's current stream waits for it (an event), the transfer is ordered on
    # torch's stream, and libdge's stream waits for that stream before it unpacks — the host waits for nothing inside an episode and runs ahead of the device, so the
    # next episode's launches (its first sort runs on a second stream and needs neither table) are queued while this episode's last kernels and the transfer are
    # still in flight.  Models without the stream-ordered entry points (the numpy stand-ins of the CPU tests) keep the blocking pair.
    ordered = hasattr(model, "export_partition_async") and (getattr(part_buf, "is_cuda", False) or getattr(model, "stream_ordered", False))
    for e in range(world):
        tgt = (rank + e) % world
        model.set_partition(world, rank, tgt)
        train_fn()
        for table in tables:
            if ordered:
                model.export_partition_async(table, world, tgt, part_buf, _stream_of(part_buf))      # what this rank just trained ...
                _ring_pass(part_buf, recv_buf, world, rank, d, transport)                            # ... goes to rank-1; rank+1's arrives:
                model.import_partition_async(table, world, (rank + 1 + e) % world, recv_buf, _stream_of(recv_buf))      # the partition of the NEXT episode
            else:
                model.export_partition(table, world, tgt, part_buf)
                _ring_pass(part_buf, recv_buf, world, rank, d, transport)
                model.import_partition(table, world, (rank + 1 + e) % world, recv_buf)
    model.set_partition(1)
    return part_buf, recv_buf


def gather_table(model, table, world, rank, part_buf=None, gather_buf=None, dist_mod=None):
    """After training: every rank holds the current partition `rank` of `table` (0 = syn0, 1 = syn1neg, 2 = syn1 under use_hs); collect the others."""
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    d = dist_mod or dist
    pf = model.partition_floats(world)
    dev = getattr(model, "torch_device", None) or "cpu"
    if part_buf is None or part_buf.numel() != pf:
        part_buf = torch.empty(pf, dtype=torch.float32, device=dev)
    if gather_buf is None or gather_buf.numel() != pf * world:
        gather_buf = torch.empty(pf * world, dtype=torch.float32, device=dev)
    model.export_partition(table, world, rank, part_buf)
    d.all_gather_into_tensor(gather_buf, part_buf)
    _wait_device(gather_buf)
    for r in range(world):
        if r != rank:
            model.import_partition(table, world, r, gather_buf[r * pf:(r + 1) * pf])


def fit_distributed(graph, n_walks, walk_len, cfg, world, rank, walk_seed, batch_walks=None, device=None, dist_mod=None):
    """w2v.fit() (J/DeepWalk.java:79) on `world` GPUs: vocabulary from the whole epoch corpus, then cfg.epochs passes of the
    block schedule over batches of `batch_walks` walks.  `graph` is this rank's replica (DeviceGraph with alias tables built);
    walks are sampled with the strided RNG, so walk i is the same on every rank and for every world size.  Returns the
    SgnsModel with complete tables on every rank."""
    import torch
    from .engine import SgnsModel
    dev = graph.device if device is None else device
    tdev = "cuda:%d" % int(dev)
    # vocabulary: every rank counts a contiguous share of the epoch's walks, the counts are summed
    lo, hi = n_walks * rank // world, n_walks * (rank + 1) // world
    counts = torch.zeros(cfg.n_vertices, dtype=torch.int64, device=tdev)
    if hi > lo:
        share = graph.sample_walks_device(hi - lo, walk_len, seed=walk_seed, rng_mode=1, first_index=lo)
        share.count_tokens(cfg.n_vertices, counts)
        share.close()
    allreduce_counts(counts, dist_mod)
    model = SgnsModel.create(cfg, counts, dev)
    nb = int(batch_walks or max(1, n_walks // 10))
    corpus = graph.sample_walks_device(min(nb, n_walks), walk_len, seed=walk_seed, rng_mode=1, first_index=0)
    bufs = (None, None)
    import torch.distributed as _dist
    transport = RingTransport.choose(world, rank, dist_mod or _dist, device=tdev) if world > 1 else None     # once, collectively
    for ep in range(cfg.epochs):
        words_before = 0
        for b0 in range(0, n_walks, nb):
            n = min(nb, n_walks - b0)
            if b0 or ep:
                graph.sample_walks_into(corpus, 0, n, walk_seed, b0)
            model.reset_stats()
            bufs = block_schedule_step(model, lambda: model.train(corpus, 0, n, walk_index_base=b0, epoch=ep, words_before=words_before,
                                                                  total_walks=n_walks), world, rank, *bufs, dist_mod=dist_mod, transport=transport)
            words_before += model.stats()["words"]
    gather_table(model, 0, world, rank, dist_mod=dist_mod)
    gather_table(model, 1, world, rank, dist_mod=dist_mod)
    if getattr(cfg, "use_hs", 0):
        gather_table(model, 2, world, rank, dist_mod=dist_mod)
    corpus.close()
    return model

