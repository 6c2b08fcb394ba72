"""Row-sharded multi-GPU execution: one process per GPU, torch.distributed (RCCL over xGMI on
MI355X, gloo on CPU for tests).

The tasks (SPARTA rows) are independent (psfrec.py:1082-1083), so the data path has no
collective.  Two small exchanges reproduce the reference's outputs:
  * all-gather of the per-task fit tables  [ntask][nl][NFIT] float64  (FIT_ROWS), and
  * sum-reduce of the per-rank partial stamp sums [nl][40][40] float64, because PSF_MEAN / FIT_MEAN
    are computed from the mean stamp over *all* tasks (psfrec.py:1104-1105).
Both are < 1 MB: latency-bound, one call each -- or ONE call for both (ShardExchange.packed: the fit
table and the stamp sum side by side in one all-gather, the stamp sums added locally); bench.py
uses the two-call form by default, the faster of the two where one GPU can measure them.
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


class ShardExchange:
    """The two exchanges of a row-sharded run with every buffer allocated once: the padded send
    block, the gathered table and (for ragged shards) the compacted table are reused by every
    step, so a step issues exactly one all-gather and one reduce and no allocation."""

    def __init__(self, ntask, nl, nfit, device, group=None, dtype=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ntask = int(ntask)
        self.bounds = shard_bounds(ntask, self.world)
        self.nmax = max(1, max(b - a for a, b in self.bounds))
        self.ragged = any(b - a != self.nmax for a, b in self.bounds)
        dtype = dtype or torch.float64
        self.buf = torch.empty((self.world * self.nmax, nl, nfit), dtype=dtype, device=device)
        self.pad = torch.zeros((self.nmax, nl, nfit), dtype=dtype, device=device) if self.ragged else None
        self.out = torch.empty((self.ntask, nl, nfit), dtype=dtype, device=device) if self.ragged else None

    def gather(self, fit_local):
        """All-gather [n_local, nl, nfit] -> [ntask, nl, nfit] in global task order (a view of
        an internal buffer: valid until the next gather on this object)."""
        import torch.distributed as dist
        a, b = self.bounds[self.rank]
        assert fit_local.shape[0] == b - a, (fit_local.shape, a, b)
        if not self.ragged:
            dist.all_gather_into_tensor(self.buf, fit_local.contiguous(), group=self.group)
            return self.buf
        self.pad[:b - a].copy_(fit_local)
        dist.all_gather_into_tensor(self.buf, self.pad, group=self.group)
        for r, (s, e) in enumerate(self.bounds):
            if e > s:
                self.out[s:e].copy_(self.buf[r * self.nmax:r * self.nmax + (e - s)])
        return self.out

    # ---- one collective per step: the rank's fit table and stamp sum side by side in one send block
    def packed(self, npix=1600):
        """Allocate the packed form: `fit_view` [n_local, nl, nfit] and `psum_view` [nl, npix] are
        views into ONE send block (the library writes its device outputs straight into them), and
        exchange_packed() is a single all-gather -- the stamp sums of the ranks are added locally,
        in rank order, by every rank (one collective, one strided copy and one small sum per step)."""
        import torch
        nl, nfit = self.buf.shape[1], self.buf.shape[2]
        a, b = self.bounds[self.rank]
        self._nfit_block = self.nmax * nl * nfit
        self._npsum = nl * npix
        self.send = torch.zeros(self._nfit_block + self._npsum, dtype=self.buf.dtype, device=self.buf.device)
        self.recv_flat = torch.empty(self.world * (self._nfit_block + self._npsum), dtype=self.buf.dtype,
                                     device=self.buf.device)
        self.recv = self.recv_flat.view(self.world, -1)
        self.table = torch.empty((self.ntask, nl, nfit), dtype=self.buf.dtype, device=self.buf.device)
        self.psum_total = torch.empty(self._npsum, dtype=self.buf.dtype, device=self.buf.device)
        self.fit_view = self.send[:(b - a) * nl * nfit].view(b - a, nl, nfit)
        self.psum_view = self.send[self._nfit_block:].view(nl, npix)
        return self

    def exchange_packed(self):
        """-> (fit_all [ntask, nl, nfit], psum_total [nl, npix]); views / buffers valid until the
        next exchange on this object."""
        import torch
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.recv_flat, self.send, group=self.group)
        nl, nfit = self.buf.shape[1], self.buf.shape[2]
        fits = self.recv[:, :self._nfit_block].view(self.world, self.nmax, nl, nfit)
        if self.ragged:
            for r, (s, e) in enumerate(self.bounds):
                if e > s:
                    self.table[s:e].copy_(fits[r, :e - s])
        else:                 # one strided copy: the fit blocks of the ranks, without the stamp sums between them
            self.table.view(self.world, self.nmax, nl, nfit).copy_(fits)
        fits = self.table
        torch.sum(self.recv[:, self._nfit_block:], dim=0, out=self.psum_total)      # rank order
        return fits, self.psum_total.view(nl, -1)

    def reduce(self, psum_local, dst=None):
        """Sum the per-rank partial stamp sums in place.  dst=None: every rank gets the total."""
        import torch.distributed as dist
        if dst is None:
            dist.all_reduce(psum_local, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.reduce(psum_local, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
        return psum_local


def gather_fit_tables(fit_local, ntask, group=None):
    """All-gather ragged [n_local, nl, nfit] float64 tensors into [ntask, nl, nfit] (global task
    order = rank order of contiguous shards).  Works for CPU (gloo) and GPU (nccl/RCCL) tensors.
    One-shot form of ShardExchange.gather (allocates its buffers)."""
    ex = ShardExchange(ntask, fit_local.shape[1], fit_local.shape[2], fit_local.device, group,
                       fit_local.dtype)
    return ex.gather(fit_local)


def reduce_psf_sum(psum_local, group=None, dst=None):
    """Sum the per-rank partial stamp sums.  dst=None: all-reduce (every rank gets the total)."""
    import torch.distributed as dist
    if dst is None:
        dist.all_reduce(psum_local, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(psum_local, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return psum_local


def reconstruct_sharded(local_compute, ntask, group=None, exchange=None):
    """Run `local_compute(start, stop) -> (fit [n_local, nl, NFIT], psf_sum [nl, 40, 40])` (torch
    tensors, float64) on this rank's shard and return (fit_all [ntask, nl, NFIT], psf_mean)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    a, b = shard_bounds(ntask, world)[rank]
    fit, psum = local_compute(a, b)
    if exchange is None:
        exchange = ShardExchange(ntask, fit.shape[1], fit.shape[2], fit.device, group, fit.dtype)
    fit_all = exchange.gather(fit)
    psum = exchange.reduce(psum.clone())
    return fit_all, psum / ntask


def context_shard_compute(ctx, lbda, seeing, gl, l0, three_lgs, h=(100, 10000), wind_speed=12.0,
                          npsflin=1, masks=None, device=None):
    """`local_compute` for reconstruct_sharded on a real GPU context: the rank's rows go through
    one mpsfr_reconstruct; the fit table and the stamp sum come back as float64 tensors on
    `device` (default: CPU, for a gloo group; pass the CUDA device for RCCL)."""
    import torch
    lbda = np.asarray(lbda, dtype=float)

    def fn(a, b):
        nl = lbda.size
        if b <= a:      # empty shard: contributes nothing
            fit = torch.zeros((0, nl, 16), dtype=torch.float64)
            psum = torch.zeros((nl, 40, 40), dtype=torch.float64)
        else:
            r = ctx.reconstruct(lbda, seeing[a:b], gl[a:b], l0[a:b], three_lgs[a:b], h,
                                wind_speed=wind_speed, npsflin=npsflin, masks=masks,
                                want_psf=False)
            fit, psum = torch.from_numpy(r['fit']), torch.from_numpy(r['psf_sum'])
        if device is not None:
            fit, psum = fit.to(device), psum.to(device)
        return fit, psum
    return fn
