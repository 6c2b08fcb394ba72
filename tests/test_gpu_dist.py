"""GPU: the row-sharded N > 1 path with the real library.  A 1-GPU box cannot run two RCCL ranks
on one device, so the two ranks exchange through gloo (CPU tensors) -- the shard arithmetic, the
gather / reduce code (muse_psfr_amd/distributed.py) and bench.py's self-launch are the ones a
multi-GPU run uses."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import H, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ntask, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import muse_psfr_amd as api
    from muse_psfr_amd.distributed import context_shard_compute, reconstruct_sharded
    see, gl, l0 = api.synthetic_rows(ntask)
    three = (np.arange(ntask) % 3 == 0).astype(np.uint8)
    lb = np.linspace(465, 930, 5)
    ctx = api.Context(dim=128, pixscale=api.grid_pixscale(128), precision='mixed', device=0)
    fit_all, mean = reconstruct_sharded(context_shard_compute(ctx, lb, see, gl, l0, three, H), ntask)
    ctx.close()
    if rank == 0:
        q.put((fit_all.numpy().copy(), mean.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,ntask', [(1, 6), (2, 7), (2, 1)])
def test_sharded_reconstruction_with_the_real_context(world, ntask):
    import muse_psfr_amd as api
    port = _free_port()
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, ntask, q)) for r in range(world)]
    for p in procs:
        p.start()
    fit_all, mean = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    see, gl, l0 = api.synthetic_rows(ntask)
    three = (np.arange(ntask) % 3 == 0).astype(np.uint8)
    ctx = api.Context(dim=128, pixscale=api.grid_pixscale(128), precision='mixed')
    r = ctx.reconstruct(np.linspace(465, 930, 5), see, gl, l0, three, H)
    ctx.close()
    np.testing.assert_array_equal(fit_all, r['fit'])          # per-task results: bit for bit
    np.testing.assert_allclose(mean, r['psf_sum'] / ntask, rtol=1e-13)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment (the driver's command form)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['MPSFR_BENCH_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
                        '--warmup', '1', '--rows', '12', '--dim', '128', '--nl', '5',
                        '--f64-steps', '2', '--profile-steps', '2'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['scaling'] == 'weak'
    assert 0 < out['roofline']['frac'] <= 1
    assert out['config']['rows_total'] == 24 and out['timed_region_repeats']['count'] > 1
    # the default at N > 1: one table row-sharded over the ranks (BASELINE.json configs[2], here 31 rows)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4',
                        '--warmup', '1', '--table-rows', '31', '--dim', '128', '--nl', '5',
                        '--f64-steps', '0', '--profile-steps', '0', '--unpruned-steps', '0', '--min-seconds', '0'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][0])
    assert out['scaling'] == 'strong' and out['config']['rows_total'] == 31
    assert 'row-sharded over 2 GPUs (16/15 rows per GPU)' in out['config']['workload']
    assert abs(out['value'] - 31 * 5 * 4 / (out['ms_per_step'] * 4e-3)) < 1e-3 * out['value']
