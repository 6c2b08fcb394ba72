"""CPU: the N > 1 path (row shards + gather of fit tables + sum-reduce of stamp sums) with two
gloo ranks; the per-shard compute is a deterministic stand-in, the collectives are the real code
of muse_psfr_amd/distributed.py that bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_local(see):
    def fn(a, b):
        s = torch.tensor(see[a:b], dtype=torch.float64)
        lam = torch.arange(1, 4, dtype=torch.float64)
        fit = (s[:, None, None] * lam[None, :, None]) * torch.arange(1, 17, dtype=torch.float64)[None, None, :]
        psum = (s.sum() * lam)[:, None, None] * torch.ones((3, 40, 40), dtype=torch.float64)
        return fit, psum
    return fn


def _worker(rank, world, port, ntask, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muse_psfr_amd.distributed import reconstruct_sharded
    from muse_psfr_amd.synthetic import synthetic_rows
    see = synthetic_rows(ntask)[0]
    fit_all, mean = reconstruct_sharded(_fake_local(see), ntask)
    if rank == 0:
        q.put((fit_all.numpy(), mean.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('ntask', [8, 7, 1])
def test_two_rank_sharding_matches_single_process(ntask):
    sys.path.insert(0, ROOT)
    from muse_psfr_amd.synthetic import synthetic_rows
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ntask, q)) for r in range(2)]
    for p in procs:
        p.start()
    fit_all, mean = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    see = synthetic_rows(ntask)[0]
    fit1, psum1 = _fake_local(see)(0, ntask)
    np.testing.assert_array_equal(fit_all, fit1.numpy())
    np.testing.assert_allclose(mean, psum1.numpy() / ntask, rtol=1e-14)
