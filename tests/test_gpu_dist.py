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
    # (ragged shards, two ranks: every rank's rows of the gathered table equal its last call's fit table)
    assert out['exchange']['gathered_table_equals_the_last_call'] is True


_RCCL_ONE_RANK = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%r, RANK='0', WORLD_SIZE='1',
                  HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
import muse_psfr_amd as api
from muse_psfr_amd import NFIT
from muse_psfr_amd.distributed import ShardExchange
n, nl = 24, 5
see, gl, l0 = api.synthetic_rows(n)
lb = np.linspace(465, 930, nl)
ctx = api.Context(dim=128, pixscale=api.grid_pixscale(128), precision='mixed', device=0)
fit = torch.zeros((n, nl, NFIT), dtype=torch.float64, device=dev)
psum = torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev)
ex = ShardExchange(n, nl, NFIT, dev)
lib_stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
for it in range(3):       # the step of bench.py at N > 1: library call, stream hand-over, collectives, event back
    ctx.reconstruct_device(lb, see, gl, l0, np.zeros(n, np.uint8), (100, 10000), 12.0, 1, None, None,
                           psum.data_ptr(), fit.data_ptr())
    cur = torch.cuda.current_stream()
    cur.wait_stream(lib_stream)
    fit_all = ex.gather(fit)
    ex.reduce(psum, dst=0)
    ev = torch.cuda.Event(); ev.record(cur)
    ctx.wait_event(ev.cuda_event)
pk = ShardExchange(n, nl, NFIT, dev).packed(1600)
for it in range(3):       # the packed form bench.py uses: the library writes into the send block of ONE all-gather
    ctx.reconstruct_device(lb, see, gl, l0, np.zeros(n, np.uint8), (100, 10000), 12.0, 1, None, None,
                           pk.psum_view.data_ptr(), pk.fit_view.data_ptr())
    cur = torch.cuda.current_stream()
    cur.wait_stream(lib_stream)
    fit_pk, psum_pk = pk.exchange_packed()
    ev = torch.cuda.Event(); ev.record(cur)
    ctx.wait_event(ev.cuda_event)
dist.barrier()
torch.cuda.synchronize()
ref = ctx.reconstruct(lb, see, gl, l0, np.zeros(n, np.uint8), (100, 10000), want_psf=False)
ok = bool(np.array_equal(fit_all.cpu().numpy(), ref['fit'])) and bool(np.allclose(psum.cpu().numpy(), ref['psf_sum'], rtol=1e-13))
ok = ok and bool(np.array_equal(fit_pk.cpu().numpy(), ref['fit'])) and bool(np.allclose(psum_pk.cpu().numpy().reshape(nl, 40, 40), ref['psf_sum'], rtol=1e-13))
print(json.dumps({'ok': ok, 'backend': dist.get_backend()}))
ctx.close()
dist.destroy_process_group()
"""


def test_shard_exchange_on_rccl_with_one_rank():
    """RCCL itself, as far as one GPU allows: a one-rank `nccl` process group on the device, the
    step of bench.py at N > 1 (asynchronous library call, stream hand-over, all-gather of the fit
    table and reduce of the stamp sum on device tensors, the event back into the library) and the
    result against the plain call.  (Two ranks cannot share a device under RCCL: the N = 2...8 runs
    are the driver's.)"""
    code = _RCCL_ONE_RANK % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and line, r.stderr[-3000:]
    out = json.loads(line[-1])
    assert out['ok'] and out['backend'] == 'nccl'


def test_bench_step_with_collectives_on_rccl_one_rank():
    """bench.py under the driver's launcher form with ONE rank, backend nccl and the collectives of the
    N > 1 step forced on: the exact code the scaling run executes per step (library call on device
    buffers, all-gather + reduce on RCCL, barrier, max over ranks), on the one GPU there is."""
    env = dict(os.environ, MPSFR_BENCH_FORCE_EXCHANGE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
           '--gpus', '1', '--steps', '10', '--warmup', '2', '--prime', '20', '--rows', '40', '--dim', '128',
           '--nl', '5', '--cpu-rows', '0', '--f64-steps', '0', '--unpruned-steps', '0', '--host-steps', '0',
           '--native-steps', '0', '--profile-steps', '0']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0 and line, r.stderr[-3000:]
    out = json.loads(line[-1])
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['steps'] == 10
    # after the fence the table the collectives gathered for the LAST step equals what the library wrote
    # for it (mpsfr_stream_wait is the only thing that orders the two)
    assert out['exchange']['gathered_table_equals_the_last_call'] is True


def test_multi_context_call_selects_every_device(tmp_path):
    """mpsfr_reconstruct_multi on a node with several GPUs: one worker thread per context, each selecting
    ITS device.  A one-GPU box cannot show that directly, so an LD_PRELOAD shim (tests/helpers) makes the
    HIP runtime report four devices -- all of them the real device 0 -- and logs every hipSetDevice per
    host thread: four contexts on "devices" 0..3 must be driven by four distinct threads, each asking
    for its own id, and the table must equal the single-context one."""
    import shutil
    import subprocess
    import textwrap
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    shim = str(tmp_path / 'shim.so')
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', shim,
                           os.path.join(ROOT, 'tests', 'helpers', 'hip_device_shim.c'), '-ldl', '-lpthread'])
    log = str(tmp_path / 'setdevice.log')
    script = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        from muse_psfr_amd import synthetic_rows, grid_pixscale
        from muse_psfr_amd._lib import Context, device_count
        assert device_count() == 4
        see, gl, l0 = synthetic_rows(40)
        lb = np.linspace(500, 900, 4)
        ps = grid_pixscale(256)
        ctxs = [Context(dim=256, pixscale=ps, device=d) for d in range(4)]
        open(%r, 'w').close()                    # only the calls below are of interest
        multi = Context.reconstruct_multi(ctxs, lb, see, gl, l0, None, (100, 10000))
        one = ctxs[0].reconstruct(lb, see, gl, l0, None, (100, 10000))
        assert np.array_equal(multi['fit'], one['fit']) and np.array_equal(multi['psf'], one['psf'])
        np.testing.assert_allclose(multi['psf_sum'], one['psf_sum'], rtol=1e-13)
        for c in ctxs:
            c.close()
        print('OK')
    ''') % (ROOT, log)
    env = dict(os.environ, LD_PRELOAD=shim, MPSFR_SHIM_DEVICES='4', MPSFR_SHIM_LOG=log)
    r = subprocess.run([sys.executable, '-c', script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    by_thread = {}
    for ln in open(log):
        th, dev = ln.split()
        by_thread.setdefault(th, set()).add(int(dev))
    # (the log also holds the single-context call and the closing of the contexts on the main thread)
    assert len([1 for d in by_thread.values() if d == {1}]) == 1, by_thread
    assert len([1 for d in by_thread.values() if d == {2}]) == 1
    assert len([1 for d in by_thread.values() if d == {3}]) == 1
