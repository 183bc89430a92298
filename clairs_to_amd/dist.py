"""Multi-GPU plumbing: candidate sites shard contiguously across ranks (one process per GPU), every rank runs the
whole hot path on its shard, and the per-site outputs are gathered in rank-major (= genomic) order.

The reference has no collective anywhere: it treats <= 10 000-site chunk files as independent jobs under GNU
parallel (run_clairs_to:1230-1308, shared/param.py:21).  The only exchange step a multi-GPU run needs is this
gather of per-site outputs (64 B/site for SNV, 96 B/site for indel); there is no reduction."""
import torch
import torch.distributed as dist


def shard_range(n_sites, world, rank):
    """Contiguous, near-equal shard [lo, hi) of a sorted candidate list; keeps overlapping windows on one GPU."""
    base, rem = divmod(int(n_sites), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_site_outputs(local, n_total, group=None):
    """local: [n_local, ...] tensor of this rank's shard (shard_range order). Returns [n_total, ...] on every rank,
    in genomic order.  Works with nccl (= RCCL over xGMI) on GPU tensors and gloo on CPU tensors."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, world, r) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "local shard does not match shard_range"
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def gather_site_rows(local, group=None):
    """The exchange step of a real run (call_chunks --gather_outputs): local [n_r, ...] per-site rows of this rank's chunks (ranks hold
    contiguous runs of the chunk list, so rank-major order is the run's order), any n_r.  Returns ([sum n_r, ...] on every rank, the
    list of n_r).  One all_gather of the row counts, one of the rows padded to the largest count (nccl = RCCL over xGMI for device
    tensors, gloo for host tensors) - no reduction."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0), counts
