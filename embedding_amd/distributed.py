"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The path shards by walk index with NO data-path collective: every rank owns a contiguous range of the epoch's walk
indices (the strided RNG makes walk i the same walk on any rank).  The only exchanges are
  * once, before training: the token counts of the shards are summed so that every rank builds the same vocabulary;
  * at epoch (or step) boundaries: delta = tables - snapshot is summed over ranks and applied with scale 1/N.
The reference is single-host; this layer is new.
"""


def _wait_device(t):
    """torch.distributed enqueues collectives on torch's streams; libdge.so works on its own stream, so the host waits
    for the tensor's device before handing the pointer over."""
    if getattr(t, "is_cuda", False):
        import torch
        torch.cuda.synchronize(t.device)


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
