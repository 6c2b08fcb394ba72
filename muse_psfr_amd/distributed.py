"""Row-sharded multi-GPU execution: one process per GPU, torch.distributed (RCCL over xGMI on
MI355X, gloo on CPU for tests).

The tasks (SPARTA rows) are independent (psfrec.py:1082-1083), so the data path has no
collective.  Two small exchanges reproduce the reference's outputs:
  * all-gather of the per-task fit tables  [ntask][nl][NFIT] float64  (FIT_ROWS), and
  * sum-reduce of the per-rank partial stamp sums [nl][40][40] float64, because PSF_MEAN / FIT_MEAN
    are computed from the mean stamp over *all* tasks (psfrec.py:1104-1105).
Both are < 1 MB: latency-bound, one call each.
"""
import numpy as np


def shard_bounds(ntask, world):
    """Contiguous, balanced blocks: [(start, stop)] * world; the first ntask % world ranks get
    one extra task.  Empty shards are allowed (ntask < world)."""
    base, extra = divmod(int(ntask), int(world))
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((s, s + n))
        s += n
    return out


def gather_fit_tables(fit_local, ntask, group=None):
    """All-gather ragged [n_local, nl, nfit] float64 tensors into [ntask, nl, nfit] (global task
    order = rank order of contiguous shards).  Works for CPU (gloo) and GPU (nccl/RCCL) tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bounds = shard_bounds(ntask, world)
    nmax = max(b - a for a, b in bounds)
    nl, nfit = fit_local.shape[1], fit_local.shape[2]
    pad = torch.zeros((nmax, nl, nfit), dtype=fit_local.dtype, device=fit_local.device)
    pad[:fit_local.shape[0]] = fit_local
    buf = torch.empty((world * nmax, nl, nfit), dtype=fit_local.dtype, device=fit_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    if all(b - a == nmax for a, b in bounds):
        return buf
    parts = [buf[r * nmax:r * nmax + (b - a)] for r, (a, b) in enumerate(bounds)]
    return torch.cat(parts, dim=0)


def reduce_psf_sum(psum_local, group=None, dst=None):
    """Sum the per-rank partial stamp sums.  dst=None: all-reduce (every rank gets the total)."""
    import torch.distributed as dist
    if dst is None:
        dist.all_reduce(psum_local, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(psum_local, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return psum_local


def reconstruct_sharded(local_compute, ntask, group=None):
    """Run `local_compute(start, stop) -> (fit [n_local, nl, NFIT], psf_sum [nl, 40, 40])` (torch
    tensors, float64) on this rank's shard and return (fit_all [ntask, nl, NFIT], psf_mean)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    a, b = shard_bounds(ntask, world)[rank]
    fit, psum = local_compute(a, b)
    fit_all = gather_fit_tables(fit, ntask, group)
    psum = reduce_psf_sum(psum.clone(), group)
    return fit_all, psum / ntask
