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


def _spawn(target, world, args):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('ntask', [1000, 1001, 5])
def test_eight_rank_sharding_matches_single_process(ntask):
    """The split of BASELINE configs[2] (1000 rows over 8 ranks: 125 each), a ragged one (1001: one rank
    takes 126) and one with empty shards (5 rows over 8 ranks), on eight gloo ranks."""
    sys.path.insert(0, ROOT)
    from muse_psfr_amd.distributed import shard_bounds
    from muse_psfr_amd.synthetic import synthetic_rows
    b = shard_bounds(ntask, 8)
    assert b[0][0] == 0 and b[-1][1] == ntask and all(b[i][1] == b[i + 1][0] for i in range(7))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1 and (ntask != 1000 or sizes == [125] * 8)
    fit_all, mean = _spawn(_worker, 8, (ntask,))
    see = synthetic_rows(ntask)[0]
    fit1, psum1 = _fake_local(see)(0, ntask)
    np.testing.assert_array_equal(fit_all, fit1.numpy())
    np.testing.assert_allclose(mean, psum1.numpy() / ntask, rtol=1e-13)


def test_bench_eight_rank_rehearsal():
    """`python bench.py --gpus 8` end to end on the CPU (MPSFR_BENCH_REHEARSAL=1): the script launches
    its own eight ranks, shards configs[2], runs the exchange on gloo and prints one JSON line with the
    whole-job value, the maximum over the ranks and every rank's own time."""
    import json
    import subprocess
    env = dict(os.environ, MPSFR_BENCH_REHEARSAL='1', MPSFR_BENCH_BACKEND='gloo', OMP_NUM_THREADS='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1',
                        '--nl', '5'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['scaling'] == 'strong' and out['rehearsal'] is True
    assert out['exchange_matches_single_process'] is True
    assert '125/125/125/125/125/125/125/125' in out['config']['workload']
    rk = out['rank_ms_per_step']
    assert len(rk['all']) == 8 and rk['max'] >= rk['min'] > 0
    assert abs(out['ms_per_step'] - rk['max']) < 1e-3 + 0.05 * rk['max']
    assert abs(out['value'] - 1000 * 5 / (out['ms_per_step'] * 1e-3)) / out['value'] < 1e-2
