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


def _worker_reuse(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muse_psfr_amd.distributed import ShardExchange, shard_bounds
    ntask = 5                                   # ragged: 3 + 2
    ex = ShardExchange(ntask, 3, 16, torch.device('cpu'))
    a, b = shard_bounds(ntask, world)[rank]
    outs = []
    ptrs = set()
    for step in range(3):                       # the same buffers serve every step
        fit = torch.full((b - a, 3, 16), float(10 * step + rank), dtype=torch.float64)
        g = ex.gather(fit)
        ptrs.add(g.data_ptr())
        psum = torch.full((3, 40, 40), float(step + 1), dtype=torch.float64)
        ex.reduce(psum, dst=0)
        outs.append((g.clone().numpy(), psum.numpy().copy()))
    if rank == 0:
        q.put((outs, len(ptrs)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_buffers_are_allocated_once():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_reuse, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs, nptr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nptr == 1
    for step, (g, psum) in enumerate(outs):
        assert g.shape == (5, 3, 16)
        assert np.all(g[:3] == 10 * step) and np.all(g[3:] == 10 * step + 1)
        assert np.all(psum == 2 * (step + 1))


def _worker_packed(rank, world, port, ntask, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muse_psfr_amd.distributed import ShardExchange, shard_bounds
    from muse_psfr_amd.synthetic import synthetic_rows
    see = synthetic_rows(ntask)[0]
    a, b = shard_bounds(ntask, world)[rank]
    ex = ShardExchange(ntask, 3, 16, torch.device('cpu')).packed(1600)
    ex2 = ShardExchange(ntask, 3, 16, torch.device('cpu'))
    outs = []
    for step in range(2):         # the producer writes straight into the views of the send block
        fit, psum = _fake_local(see * (step + 1))(a, b)
        ex.fit_view.copy_(fit)
        ex.psum_view.copy_(psum.view(3, 1600))
        f, p = ex.exchange_packed()
        f2 = ex2.gather(fit)
        p2 = ex2.reduce(psum.clone())
        outs.append((f.numpy().copy(), p.numpy().copy(), f2.numpy().copy(), p2.numpy().reshape(3, 1600).copy()))
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('ntask', [6, 5, 1])
def test_packed_exchange_is_the_two_collectives_in_one(ntask):
    """ShardExchange.packed: one all-gather carries the fit table and the stamp sum; against the
    all-gather + all-reduce form and against the single-process result, even and ragged shards
    and an empty one."""
    sys.path.insert(0, ROOT)
    from muse_psfr_amd.synthetic import synthetic_rows
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, ntask, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    see = synthetic_rows(ntask)[0]
    for step, (f, p, f2, p2) in enumerate(outs):
        fit1, psum1 = _fake_local(see * (step + 1))(0, ntask)
        np.testing.assert_array_equal(f, fit1.numpy())
        np.testing.assert_array_equal(f, f2)
        np.testing.assert_allclose(p, psum1.numpy().reshape(3, 1600), rtol=1e-14)
        np.testing.assert_allclose(p, p2, rtol=1e-15)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher environment starts N ranks through
    torch.distributed.run on 127.0.0.1 and passes its arguments on."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return R()
    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '7'] and cmd[-5].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
