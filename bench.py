#!/usr/bin/env python3
"""Benchmark of the PSF-reconstruction hot path (BASELINE.json metric: PSFs/sec, one PSF = one
(SPARTA row x wavelength) final 40x40 stamp including convolutions and Moffat fit).

    python bench.py --gpus N --steps K --warmup W

Workload: N = 1: BASELINE.json configs[1] -- 100 synthetic SPARTA rows x 35 wavelengths
(465-930 nm) on a 512^2 grid, pixscale 0.2*512/1344, npsflin=1.  N > 1:
BASELINE.json configs[2] -- the same grid and wavelengths, a 1000-row table row-sharded over the N
GPUs (muse_psfr_amd.distributed.shard_bounds: strong scaling); `--rows R` instead gives every rank
R rows (weak scaling).  One step = one pass of the hot path over the rank's rows plus, for N > 1,
the RCCL all-gather of the fit tables and sum-reduce of the partial mean-PSF numerators.
With N > 1 and no launcher environment the script starts its own N ranks
(torch.distributed.run), like the reference fans out by itself (psfrec.py:1082-1083).
Rank 0 prints ONE JSON line.  See DESIGN.md section 5 for the definition of every field.
"""
import argparse
import gc
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
DOMINANT = 'otf_rowfft'      # the kernel the roofline object describes (FFT form; 'otf_mfma' on the matrix cores)
PRIME_STEPS = 1200           # untimed, before the warm-up steps (~0.3 s: the GPU reaches its sustained clocks only after tens of ms of load,
                             # longer after the CPU-baseline leg has left it idle for half a minute)
PEAK_FP32_TFLOPS = 157.3     # MI355X fp32 vector / fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16 / bf16 matrix peak (same guide)
PEAK_HBM_GBPS = 8000.0
PEAK_FP64_TFLOPS = 78.6        # fp64 vector (= fp64 matrix) peak (same guide)
PROFILE_ROUND = 'r06'          # profiles/<round>_kernel_util.json, _traffic.json, _parity_margins.json
PIPELINE_HAS_TQ = True       # FFT form of the per-wavelength stage: the sampled first-pass lines go through HBM
sys.path.insert(0, ROOT)


def _cpu_rows(args):
    """Worker: reference-shaped oracle (4 FFTs per wavelength, fp64) for one row."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import psfr_oracle as O
    lb, s, g, l, dim, ps, masks_exact = args
    tabs = O.ao_tables((100, 10000), False, 1, exact_masks=masks_exact)
    fit, psf = O.compute_psf(lb, s, g, l, 1, (100, 10000), False, dim=dim, pixscale=ps,
                             tables=tabs)
    return fit, psf


def cpu_baseline(lb, see, gl, l0, dim, ps, nrows, cores):
    """Time the CPU oracle (kind 'port': NumPy restatement of the reference, one process per
    core over rows like the reference's joblib fan-out, psfrec.py:1082-1083) on `nrows` rows,
    and on one row in this process alone (the reference's n_jobs=1)."""
    import multiprocessing as mp
    jobs = [(lb, see[i], gl[i], l0[i], dim, ps, True) for i in range(nrows)]
    ctx = mp.get_context('fork')
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_rows, jobs[:1])      # warm imports
        t = time.time()
        res = pool.map(_cpu_rows, jobs, chunksize=1)
        dt = time.time() - t
    t = time.time()
    _cpu_rows(jobs[0])
    dt1 = time.time() - t
    fits = np.array([r[0] for r in res])
    return dict(value=nrows * lb.size / dt, seconds=dt, value_1core=lb.size / dt1), fits


def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, before this
    process touches the GPU, and leave with their exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def fft_flops(dim):
    """Nominal flops of one complex N-point transform (5 N log2 N)."""
    return 5.0 * dim * math.log2(dim)


def series_form(dim, npsflin):
    """The library's automatic choice of stage A (mpsfr_set_option "stage_a" = 1)."""
    sa = os.environ.get('MPSFR_STAGE_A')
    if sa is not None and int(float(sa)) != 1:
        return int(float(sa)) == 2
    return dim >= 512 or (dim >= 256 and npsflin >= 2)


def support_piece_fraction(dim):
    """Fraction of the half plane stage A's series form evaluates and stores (round 6): the pieces of L =
    series_lanes(N) columns of a line that touch the support of the telescope OTF -- the autocorrelation of a pupil of
    diameter N/2: the disc u^2 + y^2 < (N/2)^2 around the origin of the (wrapped) u axis."""
    L = 16 if dim <= 256 else (32 if dim == 512 else 64)
    H1, q = dim // 2 + 1, dim // L
    kept = 0
    for y in range(H1):
        w2 = (dim / 2) ** 2 - y * y
        if w2 <= 0:
            kept += 0
            continue
        w = w2 ** 0.5
        for k1 in range(q):
            lo, hi = L * k1, L * k1 + L - 1
            # columns u in [lo, hi] (u and u - N are the same frequency): inside if min(|u|, |u - N|) < w somewhere
            dmin = min(lo, dim - hi) if lo > 0 else 0
            kept += 1 if dmin < w else 0
    return kept / (H1 * q)


def hbm_model_bytes(dim, nl, rows, ndir, mixed, kept_frac=1.0, has_tq=True, series=False):
    """Algorithmic HBM bytes of one step of the pipeline AS BUILT (DESIGN.md sections 3, 5):
    every intermediate written once and read once by the next kernel, inputs/outputs once."""
    H1, NR = dim // 2 + 1, dim // 2 + 40
    p = 4 if mixed else 8
    td = rows * ndir
    psf = rows * nl
    b = {}
    if series:
        b['P (corrected-zone patch 80 x 80 fp64, write + read)'] = 2 * td * 80 * 80 * 8
        b['T (row transforms of the patch, 80 x (N/2+1) complex fp64, write + read)'] = 2 * td * 16 * 80 * H1
        b['series coefficients (one read per launch)'] = H1 * dim * (16 if mixed else 64)
        # (written only inside the telescope OTF's support, read in full tiles by the per-wavelength stage)
        b['D_phi0 (write inside the support of the telescope OTF + read)'] = int((1 + support_piece_fraction(dim)) * td * p * H1 * dim)
        b['block minima per line (write + read)'] = 2 * td * H1 * (dim // 32) * 4
    else:
        b['C (fp64 row transforms of the PSD, write + read)'] = 2 * td * 16 * H1 * NR
        b['D_phi0 (write + read)'] = 2 * td * p * H1 * dim
        if mixed and not has_tq:
            b['D_phi0 (read once more: minima for the pruning)'] = td * p * H1 * dim
    if PIPELINE_HAS_TQ and has_tq:
        b['Tq (sampled first-pass lines that survive the pruning, write + read)'] = int(
            2 * psf * 2 * p * 21 * H1 * kept_frac)
    b['stamps before the convolutions (write + read)'] = 2 * psf * 1600 * p
    b['final stamps (write, read by the fit, read by the stamp sum)'] = 3 * psf * 1600 * (4 if mixed else 8)
    b['fit table + stamp sum (write)'] = psf * 16 * 8 + nl * 1600 * 8
    return b


def rank_times_block(rank_dt, steps):
    """Every rank's own time for the timed region (`ms_per_step` is the maximum): an imbalance between
    the shards is visible in the scaling record."""
    ms = [d / steps * 1e3 for d in rank_dt]
    return {'min': round(min(ms), 4), 'max': round(max(ms), 4), 'all': [round(m, 4) for m in ms]}


def rehearse(a, rank, world, bounds, total_rows, strong):
    """MPSFR_BENCH_REHEARSAL=1: the N > 1 path of this script WITHOUT a GPU -- launcher environment, row
    shards, the exchange of muse_psfr_amd/distributed.py (all-gather of the fit tables, sum-reduce of the
    stamp sums) on gloo with CPU tensors, barrier + maximum over the ranks, the JSON line -- around a
    deterministic stand-in for the rank's rows.  What an 8-GPU node will run for the first time is then
    only the library call and RCCL itself (tests/test_dist.py runs this with 8 ranks)."""
    import torch
    import torch.distributed as dist
    from muse_psfr_amd import synthetic_rows
    from muse_psfr_amd.distributed import ShardExchange
    nl, nfit = a.nl, 16
    dist.init_process_group('gloo')
    see = synthetic_rows(total_rows)[0]
    lo, hi = bounds[rank]
    lam = torch.arange(1, nl + 1, dtype=torch.float64)
    ex = ShardExchange(total_rows, nl, nfit, torch.device('cpu'))

    def step():
        sr = torch.tensor(see[lo:hi], dtype=torch.float64)
        fit = (sr[:, None, None] * lam[None, :, None]) * torch.arange(1, nfit + 1, dtype=torch.float64)[None, None, :]
        psum = (sr.sum() * lam)[:, None, None] * torch.ones((nl, 40, 40), dtype=torch.float64)
        return ex.gather(fit), ex.reduce(psum, dst=0)
    for _ in range(a.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fit_all, psum = step()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    allt = [torch.zeros_like(tt) for _ in range(world)]
    dist.all_gather(allt, tt)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ok = True
    if rank == 0:
        sa = torch.tensor(see, dtype=torch.float64)
        want = (sa[:, None, None] * lam[None, :, None]) * torch.arange(1, nfit + 1, dtype=torch.float64)[None, None, :]
        ok = bool(torch.equal(fit_all, want)) and bool(torch.allclose(psum[:, 0, 0], sa.sum() * lam, rtol=1e-13))
        print(json.dumps({
            'metric': 'PSFs/sec (row x lambda) on %d^2 grid, %d lambda' % (a.dim, nl), 'rehearsal': True,
            'value': round(total_rows * nl * a.steps / dt, 1), 'unit': 'PSFs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'stand-in', 'data': 'synthetic',
            'config': {'workload': '%d-row synthetic SPARTA table row-sharded over %d ranks (%s rows per rank), '
                                   'stand-in compute on the CPU' % (total_rows, world, '/'.join(str(b - a_) for a_, b in bounds)),
                       'rows_total': total_rows, 'parallelism': 'rows sharded x%d' % world},
            'rank_ms_per_step': rank_times_block([float(t.item()) for t in allt], a.steps),
            'exchange_matches_single_process': ok}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def e2e_sparta_leg(ncalls):
    """compute_psf_from_sparta (psfrec.py:981-1120) end to end on synthetic SPARTA tables held in memory: 1000 rows
    x 35 wavelengths on the 512^2 grid (configs[2] on one GPU) and 100 rows on the reference's own 1280^2 grid.
    A call = HDUList in -> HDUList out (row filter, tasks, the GPU batch, FIT_ROWS / FIT_MEAN / PSF_MEAN).
    Median of `ncalls` calls after ten warm ones; `value` is the 512^2 figure."""
    import muse_psfr_amd as M
    from muse_psfr_amd import _minifits as mf
    from muse_psfr_amd.psfrec import _astropy
    fits, _ = _astropy()

    def run(nrows, dim_, ncall):
        see_, gl_, l0_ = M.synthetic_rows(nrows)
        tbl = M.create_sparta_table(nlines=nrows)
        for k in range(1, 5):
            tbl.data['LGS%d_SEEING' % k][:] = see_
            tbl.data['LGS%d_TUR_GND' % k][:] = gl_
            tbl.data['LGS%d_L0' % k][:] = l0_
        mk = (lambda: fits.HDUList([fits.PrimaryHDU(), tbl])) if fits is not None else \
             (lambda: mf.HDUList([mf.PrimaryHDU(), tbl]))
        kw = dict(verbose=False, cutoff_masks='exact')
        if dim_ != 1280:
            kw.update(dim=dim_, pixscale=M.grid_pixscale(dim_), lmin=465, lmax=930)
        for _ in range(10):                  # (context, tables, and the GPU's clocks)
            res = M.compute_psf_from_sparta(mk(), **kw)
        ts = []
        for _ in range(ncall):
            t0 = time.perf_counter()
            res = M.compute_psf_from_sparta(mk(), **kw)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        assert len(res['FIT_ROWS'].data) == nrows * 35
        return {'rows': nrows, 'dim': dim_, 'nl': 35, 'calls': ncall, 'ms_per_call_median': round(med * 1e3, 4),
                'ms_per_call_min': round(min(ts) * 1e3, 4), 'ms_per_call_max': round(max(ts) * 1e3, 4),
                'ms_per_call_p10_p90': [round(float(np.percentile(ts, 10)) * 1e3, 4), round(float(np.percentile(ts, 90)) * 1e3, 4)],
                'value': round(nrows * 35 / med, 1), 'value_min': round(nrows * 35 / max(ts), 1),
                'value_max': round(nrows * 35 / min(ts), 1)}
    a512 = run(1000, 512, ncalls)
    a1280 = run(100, 1280, ncalls)
    return {'value': a512['value'], 'unit': 'PSFs/sec', 'table_1000_rows_512': a512, 'table_100_rows_native1280': a1280,
            'astropy': fits is not None,
            'what': 'compute_psf_from_sparta(HDUList) -> HDUList with FIT_ROWS, FIT_MEAN, PSF_MEAN (psfrec.py:981-1120): '
                    'host logic, asynchronous host-output parts, table assembly and the FIT_MEAN refit included; '
                    'median wall time per call'}


LEG_REGIONS = 5        # timed regions per secondary leg: `value` = their median, min / max beside it


def spread(values, npsf_per_region=None, seconds=None):
    """Median / min / max of a leg's regions.  `values`: PSFs/s per region."""
    v = [float(x) for x in values]
    d = {'regions': len(v), 'value_median': round(float(np.median(v)), 1), 'value_min': round(min(v), 1),
         'value_max': round(max(v), 1), 'value_first': round(v[0], 1),
         'note': '`value` of this leg is the median of its regions (each bracketed by synchronisations like the main one)'}
    if seconds is not None:
        d['seconds_each'] = round(float(np.median(seconds)), 4)
    return d


def series_flops_per_line(dim):
    """Algorithmic fp64 flops of one line (td, y) of K_DPHI_SERIES: the fold of the 80
    complex inputs with the lanes' twiddles (80 complex multiply-adds x L lanes) and the Q-point in-lane
    transforms (5 Q log2 Q / 2: real outputs only) over L lanes; the polynomial, the conversions and the
    block minima are not counted."""
    L = 16 if dim <= 256 else (32 if dim == 512 else 64)
    Q = dim // L
    return 80 * 8 * L + 2.5 * Q * math.log2(Q) * L


def load_json(name):
    f = os.path.join(ROOT, 'profiles', name)
    return json.load(open(f)) if os.path.exists(f) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--prime', type=int, default=PRIME_STEPS,
                    help='untimed priming steps before the warm-up steps (profiling runs use fewer)')
    ap.add_argument('--rows', type=int, default=0,
                    help='rows per GPU per step (weak scaling); default: 100 at N = 1 (configs[1]), a 1000-row '
                         'table sharded over the ranks at N > 1 (configs[2])')
    ap.add_argument('--table-rows', type=int, default=1000, help='rows of the sharded table at N > 1')
    ap.add_argument('--dim', type=int, default=512)
    ap.add_argument('--nl', type=int, default=35)
    ap.add_argument('--npsflin', type=int, default=1)
    ap.add_argument('--precision', default='mixed', choices=['mixed', 'f64'])
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--fast-exp', type=int, default=1)
    ap.add_argument('--streams', type=int, default=0, help='pipeline lanes (0: library default)')
    ap.add_argument('--prune-eps', type=float, default=-1.0,
                    help='line pruning of the per-wavelength stage (-1: library default 1e-9, 0: off)')
    ap.add_argument('--inflight', type=int, default=1,
                    help='contexts fed in turn (each context already pipelines consecutive calls '
                         'over its two internal lanes)')
    ap.add_argument('--cpu-rows', type=int, default=-1,
                    help='rows of the CPU-baseline sample (-1: automatic, 0: skip)')
    ap.add_argument('--f64-steps', type=int, default=-1,
                    help='steps of the fp64-mode leg (-1: max(50, steps/8), 0: skip)')
    ap.add_argument('--unpruned-steps', type=int, default=-1,
                    help='steps of the leg without line pruning (-1: max(50, steps/4), 0: skip)')
    ap.add_argument('--host-steps', type=int, default=-1,
                    help='steps of the host-output leg (synchronous call, fit table + stamp sum copied to the '
                         'host; -1: 50, 0: skip)')
    ap.add_argument('--e2e-steps', type=int, default=-1,
                    help='calls of the end-to-end leg through compute_psf_from_sparta (-1: 20, 0: skip)')
    ap.add_argument('--native-steps', type=int, default=-1,
                    help='steps of the native-grid leg (1280^2, pixscale 0.2, 490-930 nm: what the reference\'s '
                         'compute_psf runs; -1: 60 at the default workload, 0: skip)')
    ap.add_argument('--min-seconds', type=float, default=0.2,
                    help='if the timed region is shorter, repeat it (value stays the first, value_min/max '
                         'report the spread)')
    ap.add_argument('--profile-every', type=int, default=4,
                    help='HIP events on every n-th launch of the dominant kernel in the timed region')
    ap.add_argument('--profile-steps', type=int, default=-1,
                    help='steps of the untimed per-kernel event pass (-1: min(steps, 40), 0: skip)')
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    a.gpus = world

    from muse_psfr_amd import synthetic_rows, grid_pixscale
    from muse_psfr_amd.distributed import shard_bounds
    dim, nl = a.dim, a.nl
    ps = grid_pixscale(dim)
    lb = np.linspace(465.0, 930.0, nl) if dim != 1280 else np.linspace(490.0, 930.0, nl)
    # the table and this rank's contiguous shard of it
    # MPSFR_BENCH_FORCE_EXCHANGE=1 (under a launcher, any N): the collectives of the N > 1 step also run
    # with one rank -- the RCCL path of this script on a box with a single GPU (tests/test_gpu_dist.py)
    xchg = world > 1 or (bool(os.environ.get('MPSFR_BENCH_FORCE_EXCHANGE')) and 'WORLD_SIZE' in os.environ)
    strong = world > 1 and a.rows <= 0
    if strong:
        total_rows = a.table_rows
        bounds = shard_bounds(total_rows, world)
    else:
        per = a.rows if a.rows > 0 else 100
        total_rows = per * world
        bounds = [(r * per, (r + 1) * per) for r in range(world)]
    see, gl, l0 = synthetic_rows(total_rows)
    sl = slice(*bounds[rank])
    rows = bounds[rank][1] - bounds[rank][0]
    if os.environ.get('MPSFR_BENCH_REHEARSAL'):
        sys.exit(rehearse(a, rank, world, bounds, total_rows, strong))
    default_workload = (world == 1 and (rows, nl, dim, a.npsflin, a.precision) == (100, 35, 512, 1, 'mixed'))
    mixed = a.precision == 'mixed'

    # ---- CPU baseline first (fork pool, before this process touches the GPU)
    cpu = None
    cpu_fits = None
    if rank == 0 and world == 1 and a.cpu_rows != 0 and a.npsflin == 1:
        # a 1-GPU box has a 16-core CPU share whatever os.cpu_count() says; override with
        # MPSFR_BENCH_CORES
        cores = int(os.environ.get('MPSFR_BENCH_CORES', min(os.cpu_count() or 1, 16)))
        # ~10-20 s of CPU work at 512^2 (about 25 PSFs/s per core)
        ncpu = min(rows, 6 * cores) if a.cpu_rows < 0 else min(rows, a.cpu_rows)
        r, cpu_fits = cpu_baseline(lb, see, gl, l0, dim, ps, ncpu, cores)
        cpu = dict(value=round(r['value'], 3), unit='PSFs/sec', cores=cores, kind='port',
                   sample='%d rows x %d lambda on %d^2 (first rows of the GPU workload), '
                          'oracle/psfr_oracle.py reference-shaped (4 FFTs/lambda, fp64, scipy '
                          'leastsq fit), one process per core, %.1f s wall' % (
                              ncpu, nl, dim, r['seconds']),
                   value_1core=round(r['value_1core'], 3))
        cal = load_json('r02_cpu_calibration.json')
        if cal:
            cpu['calibration'] = cal

    # ---- oracle sample for the native-grid leg (1280^2, pixscale 0.2: what the reference's compute_psf runs)
    nnat = (60 if default_workload else 0) if a.native_steps < 0 else a.native_steps
    if world > 1 or not mixed:
        nnat = 0
    lb_nat = np.linspace(490.0, 930.0, nl)
    nat_sel = [0, nl // 2, nl - 1]
    nat_fits = None
    if nnat > 0 and rank == 0 and a.cpu_rows != 0:
        import multiprocessing as mp
        jobs = [(lb_nat[nat_sel], see[i], gl[i], l0[i], 1280, 0.2, True) for i in range(min(4, rows))]
        with mp.get_context('fork').Pool(min(len(jobs), 4)) as pool:
            nat_fits = np.array([r[0] for r in pool.map(_cpu_rows, jobs, chunksize=1)])

    import torch
    import torch.distributed as dist
    from muse_psfr_amd import Context, NFIT
    from muse_psfr_amd._lib import load as load_lib
    # one rank per GPU; the modulo only matters when rehearsing N > 1 on a box with fewer GPUs
    # (MPSFR_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device)
    local = local % torch.cuda.device_count()
    backend = os.environ.get('MPSFR_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if xchg:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    three = np.zeros(rows, np.uint8)
    h = (100, 10000)

    from muse_psfr_amd.distributed import ShardExchange

    def make_runner(precision, nctx, prune_eps=a.prune_eps, streams=a.streams, dim=dim, ps=ps, lb=lb):
        """Steps are independent batches, pipelined through `nctx` contexts (each with its own
        HIP stream and workspaces) fed in turn.  A step is still one mpsfr_reconstruct of the
        rank's rows; every step's outputs are produced."""
        ctxs = []
        for _ in range(nctx):
            c = Context(dim=dim, pixscale=ps, precision=precision, device=local)
            if a.chunk:
                c.set_option('chunk_tasks', a.chunk)
            c.set_option('fast_exp', a.fast_exp)
            if streams:
                c.set_option('streams', streams)
            if prune_eps >= 0 and precision == 'mixed':
                c.set_option('prune_eps', prune_eps)
            if os.environ.get('MPSFR_OTF_MFMA') and precision == 'mixed':     # experiments: 0 = FFT path
                c.set_option('otf_mfma', int(os.environ['MPSFR_OTF_MFMA']))
            for key in ('mf_floor', 'mf_kernel', 'mf_permax', 'mf_mid_log2', 'mf_floor_log2', 'tier_eps', 'cold_stagger', 'stage_a', 'cu_partition', 'param_copy', 'persist_reserve', 'persist_reserve_mf', 'persist_reserve_a', 'head_fusion', 'copy_fusion', 'support_skip', 'finish_fusion', 'stage_a_queue'):      # experiments
                if os.environ.get('MPSFR_' + key.upper()) and (precision == 'mixed' or key == 'stage_a'):
                    c.set_option(key, float(os.environ['MPSFR_' + key.upper()]))
            if os.environ.get('MPSFR_PRUNE_FIXED') and precision == 'mixed':
                c.set_option('prune_fixed', int(os.environ['MPSFR_PRUNE_FIXED']))
            ctxs.append(c)
        # Two sets of result buffers per context: consecutive calls of one context overlap on its
        # internal lanes, and calls that share an output buffer would be serialised.
        nset = 2 * nctx
        # the exchange buffers are allocated once (one set per result-buffer set)
        xdev = dev if backend == 'nccl' else torch.device('cpu')
        exch = [ShardExchange(total_rows, nl, NFIT, xdev) for _ in range(nset)] if xchg else None
        packed = xchg and backend == 'nccl' and bool(os.environ.get('MPSFR_BENCH_PACKED_EXCHANGE'))
        if packed:
            # optional (MPSFR_BENCH_PACKED_EXCHANGE=1): the library writes the rank's fit table and stamp
            # sum side by side into the send block of ONE all-gather per step and the stamp sums of the
            # ranks are added locally.  Not the default: on one rank it is the slower of the two forms
            # (125 rows: 13.44 against 13.85 M PSFs/s)
            for e in exch:
                e.packed(40 * 40)
            fits = [e.fit_view for e in exch]
            psums = [e.psum_view.view(nl, 40, 40) for e in exch]
        else:
            fits = [torch.zeros((rows, nl, NFIT), dtype=torch.float64, device=dev) for _ in range(nset)]
            psums = [torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev) for _ in range(nset)]
        state = {'i': 0, 'ev': [None] * nset}
        # torch orders its collectives against the library on the GPU (no host sync inside a
        # step): torch's stream waits for the context's stream (which is ordered after every call
        # made so far) before the exchange, and the call that next overwrites a buffer set waits
        # for the exchange that last read it (mpsfr_wait_event).
        # (only a run with an exchange orders its own stream behind the library's: asking for the stream makes the
        # library join its lanes into it on every call)
        # MPSFR_BENCH_JOIN_STREAM=1: the round-4 form (torch's stream waits on the library's stream, which makes
        # every call join its lanes into it); default: mpsfr_stream_wait -- the calls stay on their lanes
        lib_streams = ([torch.cuda.ExternalStream(c.stream_handle(), device=dev) for c in ctxs]
                       if xchg and os.environ.get('MPSFR_BENCH_JOIN_STREAM') else None)

        def hand_over(k, cur):
            if lib_streams is not None:
                cur.wait_stream(lib_streams[k])
            else:
                ctxs[k].stream_wait(cur.cuda_stream)

        def step():
            k = state['i'] % nctx
            b = state['i'] % nset
            state['i'] += 1
            fit_b, psum_b = fits[b], psums[b]
            if state['ev'][b] is not None:
                ctxs[k].wait_event(state['ev'][b].cuda_event)
            ctxs[k].reconstruct_device(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin,
                                       None, None, psum_b.data_ptr(), fit_b.data_ptr())
            if xchg:           # FIT_ROWS gather + PSF_MEAN numerator reduce
                cur = torch.cuda.current_stream()
                hand_over(k, cur)
                if packed:
                    state['fit_all'], state['psum'] = exch[b].exchange_packed()
                elif backend == 'nccl':      # all-gather of the fit tables + reduce of the stamp sums
                    state['fit_all'] = exch[b].gather(fit_b)
                    exch[b].reduce(psum_b, dst=0)
                else:          # CPU rehearsal of the exchange in its two-collective form
                    state['fit_all'] = exch[b].gather(fit_b.cpu())
                    state['psum'] = exch[b].reduce(psum_b.cpu(), dst=0)
                ev = torch.cuda.Event()
                ev.record(cur)
                state['ev'][b] = ev

        def fence():
            torch.cuda.synchronize()
            for c in ctxs:
                c.sync()
            if xchg:
                dist.barrier()
            torch.cuda.synchronize()

        def timed(nsteps):
            fence()
            for c in ctxs:
                c.profile_reset()
            t0 = time.perf_counter()
            c0 = time.thread_time()
            for _ in range(nsteps):
                step()
            # wall time of the queueing loop: the GPU's pace once the host is four calls ahead (the
            # host cost proper is host_library_ms_per_call, measured inside the library)
            t_enq = time.thread_time() - c0
            fence()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            if xchg:
                # every rank's own time travels too (an imbalance between the shards shows in SCALE_r*.json)
                allt = [torch.zeros_like(tt) for _ in range(world)]
                dist.all_gather(allt, tt)
                state['rank_dt'] = [float(t.item()) for t in allt]
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()), t_enq

        def profile_sum():
            tot = {}
            for c in ctxs:
                for k, (ms, n) in c.profile().items():
                    t = tot.get(k, (0.0, 0))
                    tot[k] = (t[0] + ms, t[1] + n)
            return tot

        def close():
            for c in ctxs:
                c.close()

        def rank_times():
            return state.get('rank_dt')

        def exchange_times(nsteps):
            """ms per step of the two collectives alone (all-gather of the fit tables + reduce of the stamp
            sums), measured in an UNTIMED pass: HIP events on torch's stream around the exchange of every
            step (gloo rehearsal: host clock), the steps otherwise as in the timed region.  Every rank's
            mean travels to rank 0.  None without an exchange."""
            if not xchg:
                return None
            fence()
            evs, host_s = [], 0.0
            for _ in range(nsteps):
                k = state['i'] % nctx
                b = state['i'] % nset
                state['i'] += 1
                fit_b, psum_b = fits[b], psums[b]
                if state['ev'][b] is not None:
                    ctxs[k].wait_event(state['ev'][b].cuda_event)
                ctxs[k].reconstruct_device(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin,
                                           None, None, psum_b.data_ptr(), fit_b.data_ptr())
                cur = torch.cuda.current_stream()
                hand_over(k, cur)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                th = time.perf_counter()
                if packed:
                    exch[b].exchange_packed()
                elif backend == 'nccl':
                    exch[b].gather(fit_b)
                    exch[b].reduce(psum_b, dst=0)
                else:
                    exch[b].gather(fit_b.cpu())
                    exch[b].reduce(psum_b.cpu(), dst=0)
                host_s += time.perf_counter() - th
                e1.record(cur)
                evs.append((e0, e1))
                state['ev'][b] = e1
            fence()
            ms = (sum(a_.elapsed_time(b_) for a_, b_ in evs) if backend == 'nccl' else host_s * 1e3) / max(1, nsteps)
            tt = torch.tensor([ms], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            allt = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            return [float(t.item()) for t in allt]

        def host_lib():
            tot = [c.host_time() for c in ctxs]
            return sum(t[0] for t in tot) / max(1, sum(t[1] for t in tot))

        def exchange_check():
            """what the collectives of the LAST step gathered against what the library wrote for it.
            After the fence: this rank's rows of the gathered table equal, bit for bit, the fit table of the last
            call's buffer set, and (rank 0) the reduced stamp sum is finite and at least this rank's own.  A missed
            dependency between the library's lane and the stream of the collectives would gather stale rows.
            Every rank's verdict is combined (MIN).  None without an exchange."""
            if not xchg or 'fit_all' not in state:
                return None
            fence()
            b = (state['i'] - 1) % nset
            lo, hi = bounds[rank]
            mine = state['fit_all'][lo:hi]
            ok = bool(torch.equal(mine.cpu(), fits[b].cpu()[:hi - lo])) and bool(torch.isfinite(state['fit_all']).all())
            if rank == 0 and state.get('psum') is not None and not packed:
                ps_ = state['psum']
                ok = ok and bool(torch.isfinite(ps_).all())
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() == 1.0)
        return dict(ctxs=ctxs, step=step, fence=fence, timed=timed, profile_sum=profile_sum,
                    fits=fits, close=close, host_lib=host_lib, rank_times=rank_times, exchange_times=exchange_times,
                    exchange_check=exchange_check)

    def parity_block(fitg, n):
        return {'rows_checked': n,
                'max_abs_err_fwhm_arcsec': float(np.abs(fitg[:n, :, 5] * ps - cpu_fits[:, :, 3]).max()),
                'max_abs_err_beta': float(np.abs(fitg[:n, :, 4] - cpu_fits[:, :, 4]).max()),
                'tolerance': 1e-4}

    def kernel_times(Rx, nsteps):
        """ms per launch of every kernel slot of the library (HIP events around every launch, on the
        stream it is launched on), over `nsteps` untimed steps of the runner."""
        for c in Rx['ctxs']:
            c.set_option('profile_only', -1)
            c.set_option('profile_every', 1)
            c.set_option('profile', 1)
            c.profile_reset()
        for _ in range(nsteps):
            Rx['step']()
        Rx['fence']()
        tot = Rx['profile_sum']()
        for c in Rx['ctxs']:
            c.set_option('profile', 0)
        return {k: v[0] / v[1] for k, v in tot.items() if v[1]}, {k: v[1] / nsteps for k, v in tot.items() if v[1]}

    def probe_mf_work(ctx, lbv):
        """Tile steps of the matrix-core stage per row, from one call over the first rows."""
        nprobe = min(rows, 32)
        pf = torch.zeros((nprobe, nl, NFIT), dtype=torch.float64, device=dev)
        psm = torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev)
        sp = slice(sl.start, sl.start + nprobe)
        try:
            ctx.reconstruct_device(lbv, see[sp], gl[sp], l0[sp], three[:nprobe], h, 12.0, a.npsflin,
                                   None, None, psm.data_ptr(), pf.data_ptr())
            ctx.sync()
            w = ctx.debug_fetch('mf_work', (7,))
        except Exception:
            return None
        return {'tile_steps_per_row': w[0] / nprobe, 'tiles_per_row': w[1] / nprobe,
                'tile_steps_unpruned_per_row': w[2] / nprobe,
                'full_steps_per_row': w[3] / nprobe, 'mid_steps_per_row': w[4] / nprobe,
                'union_steps_per_row': w[5] / nprobe, 'support_steps_per_row': w[6] / nprobe}

    def mfma_roofline(mfw, ntask_launch, ms_launch, dimv):
        """`roofline` object of the matrix-core per-wavelength stage for a launch over ntask_launch rows."""
        nmfma = (mfw['full_steps_per_row'] * 9 + mfw['mid_steps_per_row'] * 6) * ntask_launch
        tiles = mfw['tiles_per_row'] * ntask_launch
        fl = nmfma * 2 * 16 * 16 * 32 + tiles * 24 * 2 * 16 * 16 * 16
        nmt = (dimv // 2 + 1 + 15) // 16
        fl32_all = (mfw['tile_steps_unpruned_per_row'] * 2 * 16 * 32 * 42 + nmt * nl * 2 * 2 * 16 * 21 * 21) * ntask_launch
        t = ms_launch * 1e-3
        return {'bound': 'mfma', 'kernel': 'otf_mfma (k_otf_mfma2 + k_mf_finish)', 'achieved': round(fl / t / 1e12, 2),
                'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(fl / t / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                'avg_launch_ms': round(ms_launch, 4), 'executed_flops_per_launch': fl,
                'tile_steps_executed_fraction': round(mfw['tile_steps_per_row'] / mfw['tile_steps_unpruned_per_row'], 4),
                'blocks_kept_by_any_wavelength_fraction': round(float(mfw['union_steps_per_row']) * nl / mfw['tile_steps_unpruned_per_row'], 4)
                if mfw.get('union_steps_per_row', -1) >= 0 else None,
                'blocks_inside_telescope_support_fraction': round(float(mfw['support_steps_per_row']) * nl / mfw['tile_steps_unpruned_per_row'], 4)
                if mfw.get('support_steps_per_row', -1) >= 0 else None,
                'unpruned_fp32_equivalent': {
                    'flops_per_launch': fl32_all, 'rate_TFLOPs': round(fl32_all / t / 1e12, 1),
                    'over_fp32_peak': round(fl32_all / t / 1e12 / PEAK_FP32_TFLOPS, 3),
                    'note': 'the whole contraction of the launch (every block of the half plane, one product per '
                            'element, no padding) per second: work that does not shrink when the pruning improves, '
                            'so the figure only rises when the launch gets faster (> 1 x the fp32 peak = what the '
                            'pruning and the matrix cores buy over a dense fp32 evaluation)'}}

    def stage_a_roofline(ktimes, ntd_launch, dimv, mixedv):
        """fp64 vector-pipe roofline of the series form of stage A (K_DPHI_SERIES), when it ran."""
        if 'dphi_series' not in ktimes:
            return None
        lines = ntd_launch * (dimv // 2 + 1)
        fl = lines * series_flops_per_line(dimv)
        t = ktimes['dphi_series'] * 1e-3
        byt = lines * (80 * 16 + dimv * (4 if mixedv else 8) + (dimv // 32) * 4)
        return {'bound': 'valu_fp64', 'kernel': 'dphi_series (k_dphi_series)', 'achieved': round(fl / t / 1e12, 2),
                'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(fl / t / 1e12 / PEAK_FP64_TFLOPS, 4),
                'avg_launch_ms': round(ktimes['dphi_series'], 4), 'flops_per_line': series_flops_per_line(dimv),
                'lines_per_launch': lines,
                'sustained_fp64_peak_note': 'a loop of independent v_fma_f64 on every SIMD of this chip sustains 58 TFLOP/s '
                                            '(the clock falls to ~1.45 GHz under fp64 load: scripts/ubench/dpp64.hip, '
                                            'profiles/%s_ubench_dpp64.txt)' % PROFILE_ROUND,
                'hbm_GBps': round(byt / t / 1e9, 1), 'hbm_frac': round(byt / t / 1e9 / PEAK_HBM_GBPS, 4),
                'model': '80 complex multiply-adds x L lanes + in-lane Q-point transforms (real outputs) per line; '
                         'bytes: 1280 in + 4 N (8 N in f64 mode) out + N/8 of block minima per line'}

    # the steps are queued from Python: a cyclic-GC pass in the middle of the timed loop (tens of
    # ms with torch loaded) would starve the GPU, so collection is parked for the measurement
    gc.collect()
    gc.disable()
    R = make_runner(a.precision, max(1, a.inflight))
    ctxs = R['ctxs']
    # untimed priming before the W warm-up steps: lets the HIP runtime grow its command/signal
    # pools to the depth the host runs ahead by, and the GPU leave its idle clocks after the CPU
    # baseline (the first ~50 calls of a process are 5-10 % slower)
    for _ in range(a.prime):
        R['step']()
    R['fence']()
    # Timed region: HIP events only around the dominant kernel (roofline.achieved); bracketing
    # every launch costs ~8 % of a step in event packets, so the per-kernel table comes from a
    # second, untimed pass.
    # which form of the per-wavelength stage runs: one probing step with every slot timed
    for c in ctxs:
        c.set_option('profile_only', -1)
        c.set_option('profile_every', 1)
        c.set_option('profile', 1)
        c.profile_reset()
    R['step']()
    R['fence']()
    dominant = 'otf_mfma' if R['profile_sum']().get('otf_mfma', (0, 0))[1] > 0 else DOMINANT
    # (every fourth launch of it: a timed launch stops its queue in front of and behind the kernel, ~12 us in the
    # kernel trace -- profiles/r05_experiments.md; `roofline.launches` says how many were timed)
    for c in ctxs:
        c.set_option('profile_only', c.profile_names().index(dominant))
        c.set_option('profile_every', a.profile_every)
        c.set_option('profile', 1)
    # The library's pool of HIP events for the bracketed kernel must hold a whole timed region's worth
    # before that region starts (an event created inside it costs ~10 us of host time: with K = 20 and
    # W = 5 the first region lost 3-7 % to the fifteen pairs it had to create): K untimed steps in the mode
    # of the timed ones, their events handed back to the pool.
    for _ in range(a.steps):
        R['step']()
    R['fence']()
    for c in ctxs:
        c.profile_reset()
    # the W warm-up steps run in exactly the mode of the timed ones, right before them
    for _ in range(a.warmup):
        R['step']()
    dt, t_enq = R['timed'](a.steps)
    exch_ok = R['exchange_check']()
    rank_dt = R['rank_times']()
    prof = R['profile_sum']()
    host_lib_s = R['host_lib']()
    # A region of a few milliseconds is thin evidence: repeat it (same K steps, same brackets) until
    # --min-seconds of timed work have been seen; `value` stays the FIRST region, the repeats only
    # report the spread.
    rep_dt = [dt]
    while sum(rep_dt) < a.min_seconds and len(rep_dt) < 64:
        rep_dt.append(R['timed'](a.steps)[0])
    nprof = min(a.steps, 40) if a.profile_steps < 0 else a.profile_steps
    prof_all = {}
    if nprof > 0:
        for c in ctxs:
            c.set_option('profile_only', -1)
            c.set_option('profile_every', 1)
            c.profile_reset()
        for _ in range(nprof):
            R['step']()
        R['fence']()
        prof_all = R['profile_sum']()
    for c in ctxs:
        c.set_option('profile', 0)
    # N > 1: what the two collectives of a step cost on their own (untimed pass, events around them)
    exch_ms = R['exchange_times'](min(a.steps, 40))
    fitg = R['fits'][0].cpu().numpy()
    # line pruning: lines of the OTF half plane transformed per (row, wavelength pair) in the last
    # call (every call of the loop has the same inputs)
    lines_kept = None
    if mixed and a.prune_eps != 0 and not os.environ.get('MPSFR_PRUNE_FIXED'):
        # measured on one single-chunk call over the first rows of the workload
        nprobe = min(rows, 4096 // nl if nl <= 512 else 8)
        if a.chunk:
            nprobe = min(nprobe, a.chunk)
        pf = torch.zeros((nprobe, nl, NFIT), dtype=torch.float64, device=dev)
        psm = torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev)
        sp = slice(sl.start, sl.start + nprobe)
        ctxs[0].reconstruct_device(lb, see[sp], gl[sp], l0[sp], three[:nprobe], h, 12.0, a.npsflin,
                                   None, None, psm.data_ptr(), pf.data_ptr())
        ctxs[0].sync()
        try:
            lines_kept = ctxs[0].debug_fetch('vkeep', (nprobe, (nl + 1) // 2))
        except Exception:       # pruning switched off in the library
            lines_kept = None
    # matrix-core per-wavelength kernel: tile steps it executes per row (same probe call, or one
    # call over the first rows when pruning is off)
    mf_work = None
    if mixed:
        try:
            if lines_kept is None:
                nprobe = min(rows, 4096 // nl if nl <= 512 else 8)
                pf = torch.zeros((nprobe, nl, NFIT), dtype=torch.float64, device=dev)
                psm = torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev)
                sp = slice(sl.start, sl.start + nprobe)
                ctxs[0].reconstruct_device(lb, see[sp], gl[sp], l0[sp], three[:nprobe], h, 12.0, a.npsflin,
                                           None, None, psm.data_ptr(), pf.data_ptr())
                ctxs[0].sync()
            w = ctxs[0].debug_fetch('mf_work', (7,))
            mf_work = {'tile_steps_per_row': w[0] / nprobe, 'tiles_per_row': w[1] / nprobe,
                       'tile_steps_unpruned_per_row': w[2] / nprobe,
                       'full_steps_per_row': w[3] / nprobe, 'mid_steps_per_row': w[4] / nprobe,
                       'union_steps_per_row': w[5] / nprobe, 'support_steps_per_row': w[6] / nprobe}
        except Exception:       # the FFT kernels ran (several directions, or otf_mfma = 0)
            mf_work = None
    R['close']()

    def apply_options(c, precision):
        if a.chunk:
            c.set_option('chunk_tasks', a.chunk)
        if a.streams:
            c.set_option('streams', a.streams)
        if a.prune_eps >= 0 and precision == 'mixed':
            c.set_option('prune_eps', a.prune_eps)

    # ---- what a caller of compute_psf_from_sparta gets: the fit table and the stamp sum in (pageable)
    # host memory -- PCIe inclusive, never `value`.  Asynchronous host-output calls (on_device = 2), two
    # in flight: the results of call k are collected while call k + 1 runs, like the passes of a table
    # larger than one call; `sync` beside it is the blocking call, one at a time.
    host_leg = None
    nhost = (50 if a.host_steps < 0 else a.host_steps) if world == 1 else 0
    if nhost > 0:
        c = Context(dim=dim, pixscale=ps, precision=a.precision, device=local)
        apply_options(c, a.precision)
        for _ in range(3):
            c.reconstruct(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, want_psf=False)
        t0 = time.perf_counter()
        for _ in range(nhost):
            rs = c.reconstruct(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, want_psf=False)
        dts = time.perf_counter() - t0
        nh2 = 4 * nhost
        pend = []
        for _ in range(8):
            pend.append(c.reconstruct_async(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, want_psf=False))
            if len(pend) > 2:
                pend.pop(0).wait()
        while pend:
            pend.pop(0).wait()
        dths = []
        for _ in range(LEG_REGIONS):
            t0 = time.perf_counter()
            for _ in range(nh2):
                pend.append(c.reconstruct_async(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, want_psf=False))
                if len(pend) > 2:
                    rh = pend.pop(0).wait()
            while pend:
                rh = pend.pop(0).wait()
            dths.append(time.perf_counter() - t0)
        dth = float(np.median(dths))
        # (the blocking form too)
        dtss = [dts]
        for _ in range(LEG_REGIONS - 1):
            t0 = time.perf_counter()
            for _ in range(nhost):
                rs = c.reconstruct(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, want_psf=False)
            dtss.append(time.perf_counter() - t0)
        dts = float(np.median(dtss))
        c.close()
        host_leg = {'value': round(rows * nl * nh2 / dth, 1), 'unit': 'PSFs/sec', 'steps': nh2,
                    'spread': spread([rows * nl * nh2 / d for d in dths], seconds=dths),
                    'ms_per_call': round(dth / nh2 * 1e3, 4), 'calls_in_flight': 2,
                    'outputs': 'fit table [rows][nl][16] + stamp sum [nl][40][40] (float64) in host memory, '
                               'asynchronous host-output calls (on_device = 2), results collected one call behind: '
                               'what compute_psf_from_sparta hands back (psfrec.py:978, 1104-1113)',
                    'sync': {'value': round(rows * nl * nhost / dts, 1), 'ms_per_call': round(dts / nhost * 1e3, 4),
                             'steps': nhost, 'note': 'blocking calls (on_device = 0), one at a time',
                             'spread': spread([rows * nl * nhost / d for d in dtss], seconds=dtss)},
                    'async_equals_sync': bool(np.array_equal(rh['fit'], rs['fit']) and np.array_equal(rh['psf_sum'], rs['psf_sum']))}
        if cpu_fits is not None:
            n = cpu_fits.shape[0]
            host_leg['parity'] = {'rows_checked': n,
                                  'max_abs_err_fwhm_arcsec': float(np.abs(rh['fit'][:n, :, 5] * ps - cpu_fits[:, :, 3]).max()),
                                  'max_abs_err_beta': float(np.abs(rh['fit'][:n, :, 4] - cpu_fits[:, :, 4]).max()),
                                  'tolerance': 1e-4}

    # ---- the drop-in API end to end (psfrec.py:981-1120): FITS table in memory -> compute_psf_from_sparta -> HDUList
    # (FIT_ROWS, FIT_MEAN, PSF_MEAN), host logic, PCIe and table assembly included -- never `value`
    e2e = None
    ne2e = (20 if a.e2e_steps < 0 else a.e2e_steps) if (world == 1 and a.precision == 'mixed' and dim == 512 and a.npsflin == 1) else 0
    if ne2e > 0:
        e2e = e2e_sparta_leg(ne2e)

    # ---- the reference's own grid (compute_psf hard-codes dim = 1280, pixscale 0.2: psfrec.py:954-955, 659)
    native = None
    if nnat > 0:
        Rn = make_runner('mixed', 1, dim=1280, ps=0.2, lb=lb_nat)
        for _ in range(100):        # (priming: 60 ms of load, like the 1200 steps before the main region -- with 8 the
            Rn['step']()            # 20-step region came out 10 % below the same workload run on its own)
        dtns = [Rn['timed'](nnat)[0] for _ in range(LEG_REGIONS)]
        dtn = float(np.median(dtns))
        fitn = Rn['fits'][0].cpu().numpy()
        Rn['close']()
        # per-kernel times with one call in flight (one lane: nothing else on the GPU beside a kernel)
        Rn = make_runner('mixed', 1, streams=1, dim=1280, ps=0.2, lb=lb_nat)
        for _ in range(6):
            Rn['step']()
        ktn, kln = kernel_times(Rn, 8)
        mfn = probe_mf_work(Rn['ctxs'][0], lb_nat)
        Rn['close']()
        native = {'value': round(rows * nl * nnat / dtn, 1), 'unit': 'PSFs/sec', 'steps': nnat,
                  'spread': spread([rows * nl * nnat / d for d in dtns], seconds=dtns),
                  'ms_per_step': round(dtn / nnat * 1e3, 4),
                  'workload': '%d rows x %d lambda (490-930 nm), 1280^2 grid, pixscale 0.2' % (rows, nl),
                  'kernel_ms_per_launch_one_call_in_flight': {k: round(v, 4) for k, v in ktn.items()}}
        if mfn is not None and 'otf_mfma' in ktn:
            native['roofline'] = mfma_roofline(mfn, rows / kln['otf_mfma'], ktn['otf_mfma'], 1280)
            mb = hbm_model_bytes(1280, nl, rows, 1, True, has_tq=False, series=series_form(1280, 1))
            native['roofline']['hbm'] = {'model_bytes_per_step': sum(mb.values()), 'model_terms': mb,
                                         'achieved_GBps': round(sum(mb.values()) / (dtn / nnat) / 1e9, 1),
                                         'peak': PEAK_HBM_GBPS,
                                         'frac': round(sum(mb.values()) / (dtn / nnat) / 1e9 / PEAK_HBM_GBPS, 4)}
            native['roofline']['hbm_frac'] = native['roofline']['hbm']['frac']
        sa = stage_a_roofline(ktn, rows / kln.get('dphi_series', 1), 1280, True)
        if sa:
            native['roofline_stage_a'] = sa
        if nat_fits is not None:
            n = nat_fits.shape[0]
            native['parity'] = {'rows_checked': n, 'wavelengths_checked': [float(lb_nat[i]) for i in nat_sel],
                                'max_abs_err_fwhm_arcsec': float(np.abs(fitn[:n][:, nat_sel, 5] * 0.2 - nat_fits[:, :, 3]).max()),
                                'max_abs_err_beta': float(np.abs(fitn[:n][:, nat_sel, 4] - nat_fits[:, :, 4]).max()),
                                'tolerance': 1e-4}

    # ---- the dominant kernel with one call in flight (no other kernel beside it on the GPU): what
    # the kernel itself achieves, as opposed to its share of the GPU in the pipelined run
    alone_ms = None
    kt_alone, kl_alone = {}, {}
    if mixed and a.profile_steps != 0:
        R4 = make_runner(a.precision, 1, streams=1)
        for _ in range(8):
            R4['step']()
        for c in R4['ctxs']:
            c.set_option('profile_only', c.profile_names().index(dominant))
            c.set_option('profile', 1)
        R4['timed'](40)
        ms4, n4 = R4['profile_sum']()[dominant]
        alone_ms = ms4 / max(n4, 1)
        kt_alone, kl_alone = kernel_times(R4, 20)       # every kernel, one call in flight
        R4['close']()

    # ---- the same workload with every line of the half plane transformed (prune_eps = 0)
    unpruned = None
    pruned = mixed and a.prune_eps != 0 and not os.environ.get('MPSFR_PRUNE_FIXED')
    nunp = (max(50, a.steps // 4) if a.unpruned_steps < 0 else a.unpruned_steps) if pruned else 0
    if nunp > 0:
        R3 = make_runner('mixed', max(1, a.inflight), prune_eps=0.0)
        for _ in range(100):        # (priming, as for the native leg)
            R3['step']()
        dt3s = [R3['timed'](nunp)[0] for _ in range(LEG_REGIONS)]
        dt3 = float(np.median(dt3s))
        fit3 = R3['fits'][0].cpu().numpy()
        R3['close']()
        unpruned = {'value': round(total_rows * nl * nunp / dt3, 1), 'unit': 'PSFs/sec', 'steps': nunp,
                    'spread': spread([total_rows * nl * nunp / d for d in dt3s], seconds=dt3s),
                    'ms_per_step': round(dt3 / nunp * 1e3, 4),
                    'max_abs_diff_fwhm_px_vs_pruned': float(np.abs(fit3[:, :, 5] - fitg[:, :, 5]).max()),
                    'max_abs_diff_beta_vs_pruned': float(np.abs(fit3[:, :, 4] - fitg[:, :, 4]).max())}

    # ---- the same workload at the reference's own precision (fp64 everywhere), fewer steps
    f64 = None
    # (at least 50 steps: a region of a few calls is mostly the fill and drain of the pipeline)
    nf64 = (max(50, a.steps // 8) if a.f64_steps < 0 else a.f64_steps) if mixed else 0
    if nf64 > 0:
        R2 = make_runner('f64', max(1, a.inflight))
        for _ in range(100):        # (priming, as for the native leg: with 4 the leg read 4.7 M where the
            R2['step']()            # primed run of the same workload gives 4.9 M)
        dt2s = [R2['timed'](nf64)[0] for _ in range(LEG_REGIONS)]
        dt2 = float(np.median(dt2s))
        fit2 = R2['fits'][0].cpu().numpy()
        R2['close']()
        R2 = make_runner('f64', 1, streams=1)        # per-kernel times with one call in flight
        for _ in range(4):
            R2['step']()
        kt2, kl2 = kernel_times(R2, 8)
        vk2 = None
        try:
            vk2 = R2['ctxs'][0].debug_fetch('vkeep', (min(rows, 512), (nl + 1) // 2)) if rows <= 512 else None
        except Exception:
            vk2 = None
        R2['close']()
        f64 = {'value': round(total_rows * nl * nf64 / dt2, 1), 'unit': 'PSFs/sec',
               'spread': spread([total_rows * nl * nf64 / d for d in dt2s], seconds=dt2s),
               'steps': nf64, 'ms_per_step': round(dt2 / nf64 * 1e3, 4), 'dtype': 'f64',
               'kernel_ms_per_launch_one_call_in_flight': {k: round(v, 4) for k, v in kt2.items()}}
        if 'otf_rowfft' in kt2:
            # dominant kernel of the f64 mode: the fp64 line transforms of the per-wavelength stage (two
            # wavelengths per complex N-point transform, pruned lines not counted) + two exponentials per
            # OTF element and pair; bounded by the fp64 vector pipe (profiles/r04_pmc_f64.txt)
            kf = float(vk2.mean() / (dim // 2 + 1)) if vk2 is not None else 1.0
            ntr = rows / kl2['otf_rowfft'] * (dim // 2 + 1) * ((nl + 1) // 2) * kf
            fl = ntr * (fft_flops(dim) + 2 * dim * 21)
            t = kt2['otf_rowfft'] * 1e-3
            f64['roofline'] = {'bound': 'valu_fp64', 'kernel': 'otf_rowfft (k_otf_rowfft<double>)',
                               'achieved': round(fl / t / 1e12, 2), 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(fl / t / 1e12 / PEAK_FP64_TFLOPS, 4), 'avg_launch_ms': round(kt2['otf_rowfft'], 4),
                               'lines_transformed_fraction': round(kf, 4),
                               'model': 'lines transformed x (5 N log2 N + 2 N exponentials of ~21 fp64 operations)',
                               'traffic': None}
            mb = hbm_model_bytes(dim, nl, rows, a.npsflin ** 2, False, kept_frac=kf, series=series_form(dim, a.npsflin))
            f64['roofline']['hbm'] = {'model_bytes_per_step': sum(mb.values()),
                                      'achieved_GBps': round(sum(mb.values()) / (dt2 / nf64) / 1e9, 1), 'peak': PEAK_HBM_GBPS,
                                      'frac': round(sum(mb.values()) / (dt2 / nf64) / 1e9 / PEAK_HBM_GBPS, 4)}
            f64['roofline']['hbm_frac'] = f64['roofline']['hbm']['frac']
        sa2 = stage_a_roofline(kt2, rows * a.npsflin ** 2 / kl2.get('dphi_series', 1), dim, False)
        if sa2:
            f64['roofline_stage_a'] = sa2
        if cpu_fits is not None and rank == 0:
            f64['parity'] = parity_block(fit2, cpu_fits.shape[0])
    gc.enable()

    if rank == 0:
        npsf = total_rows * nl * a.steps
        ndir = a.npsflin ** 2
        ms, nlaunch = prof[dominant]
        chunk = a.chunk or 'auto'
        # launches of the kernel per step (one per pipeline pass): from the all-kernel pass, where every launch is
        # timed; in the timed region only every `--profile-every`-th launch carries events
        if prof_all.get(dominant, (0, 0))[1] and nprof:
            launches_per_step = prof_all[dominant][1] / nprof
        else:
            launches_per_step = max(1.0, round(nlaunch * a.profile_every / a.steps))
        units_per_launch = rows * nl * ndir / launches_per_step
        tasks_per_launch = rows / launches_per_step
        avg_s = ms / max(nlaunch, 1) * 1e-3
        ntrans_all = tasks_per_launch * (dim // 2 + 1) * ((nl + 1) // 2)
        kept_frac = float(lines_kept.mean() / (dim // 2 + 1)) if lines_kept is not None else 1.0
        if mf_work is not None:
            # Dominant kernel = the per-wavelength stage on the matrix cores (otf_mfma.hip).  Executed
            # arithmetic: a tile step (16 lines x 32 columns of the OTF against 48 table columns) is
            # nine v_mfma_f32_16x16x32_f16 (three fp16 products per fp32-grade product), the second
            # pass of an m-tile 24 v_mfma_f32_16x16x16_f16.  Only what the pruned launch executes is
            # counted, so the fraction of the dense fp16 matrix peak is <= 1 by construction.  The
            # fp32-equivalent figure counts one product per element and no padding (42 of 48
            # columns; 21 x 21 of 32 x 32 in the second pass): the arithmetic the FFT path's
            # fraction of the fp32 peak was quoted on.
            steps = mf_work['tile_steps_per_row'] * tasks_per_launch
            tiles = mf_work['tiles_per_row'] * tasks_per_launch
            # a full tile step is nine MFMA, a mid one (no low half of the OTF: otf_mfma2.hip) six
            nmfma = (mf_work['full_steps_per_row'] * 9 + mf_work['mid_steps_per_row'] * 6) * tasks_per_launch
            flops = nmfma * 2 * 16 * 16 * 32 + tiles * 24 * 2 * 16 * 16 * 16
            flops32 = steps * 2 * 16 * 32 * 42 + tiles * 2 * 2 * 16 * 21 * 21
            peak = PEAK_F16_MFMA_TFLOPS
            bound = 'mfma'
            kernel_name = 'otf_mfma'
            model_txt = ('tile steps executed (block pruning: 16 x 32 blocks of the OTF half plane per row and '
                         'wavelength) x 9 x v_mfma_f32_16x16x32_f16 + m-tiles x 24 x v_mfma_f32_16x16x16_f16; '
                         'HIP events on the launch stream, timed region')
        else:
            # FFT kernels (f64 mode, several directions): nominal arithmetic = the complex line
            # transforms performed, (N/2+1) lines per task, two wavelengths per complex transform,
            # 5 N log2 N flops each; pruned lines are not counted.
            ntrans = ntrans_all * kept_frac
            flops = ntrans * fft_flops(dim)
            flops32 = None
            peak = PEAK_FP32_TFLOPS if mixed else PEAK_FP32_TFLOPS / 2
            bound = 'valu_fp32' if mixed else 'valu_fp64'
            kernel_name = dominant
            model_txt = ('lines transformed (line pruning: vkeep per row and wavelength pair, of (N/2+1) x '
                         'ceil(nl/2) per row) x 5 N log2 N flops per complex N-point transform; HIP events on '
                         'the launch stream, timed region')
        achieved = flops / avg_s / 1e12 if avg_s > 0 else 0.0
        util_name = next((n for n in (PROFILE_ROUND + '_kernel_util.json',) if load_json(n)), None)
        util = load_json(util_name) if util_name else {}
        kkey = 'k_otf_mfma' if mf_work is not None else 'k_' + dominant
        u = next((v for k, v in util.items() if k.startswith(kkey)), None) if (dim, mixed) == (512, True) else None
        tj = load_json(PROFILE_ROUND + '_traffic.json')
        traffic = None
        traffic_step = None
        if tj and (dim, nl, rows, a.npsflin, mixed) == (512, 35, 100, 1, True):
            k = next((v for kk, v in tj['kernels'].items() if kk.startswith(kkey)), {})
            if k.get('fetch_kib') is not None and k.get('write_kib') is not None:
                per_unit = (k['fetch_kib'] + k['write_kib']) * 1024.0 / tj['units_per_launch']
                traffic = per_unit * units_per_launch
            traffic_step = sum((v.get('fetch_kib', 0) + v.get('write_kib', 0)) * 1024.0
                               for v in tj['kernels'].values())
        # step-level utilisation: what the kernels of ONE call keep busy, summed over the call
        # (PMC passes with one call in flight, profiles/<round>_kernel_util.json), over the pipelined step time
        step_util = None
        if util and (dim, nl, rows, a.npsflin, mixed) == (512, 35, 100, 1, True):
            ks = {k: v for k, v in util.items() if isinstance(v, dict) and 'avg_us' in v}
            tot_us = sum(v['avg_us'] for v in ks.values())
            step_us = dt / a.steps * 1e6
            busy = {key: sum(v['avg_us'] * (v.get(src) or 0.0) for v in ks.values())
                    for key, src in (('valu_issue', 'valu_issue'), ('mfma_busy', 'mfma_busy'), ('lds_busy', 'lds_array_busy'))}
            pipelined = {k: v[0] / max(nprof, 1) for k, v in prof_all.items() if v[1]}
            big = max(pipelined, key=pipelined.get) if pipelined else None
            step_util = {'kernel_us_per_call_one_in_flight': round(tot_us, 1), 'step_us': round(step_us, 1),
                         'lane_overlap': round(tot_us / step_us, 3),
                         'valu_issue_us': round(busy['valu_issue'], 1), 'mfma_busy_us': round(busy['mfma_busy'], 1),
                         'lds_busy_us': round(busy['lds_busy'], 1),
                         'valu_issue_frac': round(busy['valu_issue'] / step_us, 4),
                         'mfma_busy_frac': round(busy['mfma_busy'] / step_us, 4),
                         'lds_busy_frac': round(busy['lds_busy'] / step_us, 4),
                         'largest_pipelined_share': big and {'kernel': big, 'ms_per_step': round(pipelined[big], 4),
                                                            'share_of_kernel_time': round(pipelined[big] / sum(pipelined.values()), 4)},
                         'note': 'sum over the kernels of a call of (duration x share of cycles the unit is busy) / time per '
                                 'step of the two-lane pipeline: no unit of the chip is busy for more than this fraction of a '
                                 'step -- the path is bound by latency and occupancy (registers, LDS), not by a pipe or by HBM',
                         'source': 'profiles/%s' % util_name}
        series = 'dphi_series' in prof_all or ('dphi_series' in prof)
        model = hbm_model_bytes(dim, nl, rows, ndir, mixed, kept_frac, has_tq=mf_work is None,
                                series=series or (not prof_all and series_form(dim, a.npsflin)))
        if tj and 'model_extra' in tj:          # e.g. an intermediate the current pipeline keeps
            model.update(tj['model_extra'])
        model_step = float(sum(model.values()))
        step_s = dt / a.steps
        legacy_bytes_per_psf = ndir * dim * dim * (3 * (4 if mixed else 8) + (5 * 8 + (4 if mixed else 8)) / nl)
        out = {
            'metric': 'PSFs/sec (row x lambda) on %d^2 grid, %d lambda' % (dim, nl),
            'value': round(npsf / dt, 1), 'unit': 'PSFs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(step_s * 1e3, 4),
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': ('f64 PSD->structure function, per-lambda OTF f32 -> split-fp16 MFMA (fp32 accumulate), '
                      'f64 fit' if mf_work is not None else
                      'f64 PSD->structure function, f32 per-lambda OTF/FFT, f64 fit') if mixed else 'f64',
            'data': 'synthetic',
            'config': {'workload': ('%d-row synthetic SPARTA table row-sharded over %d GPUs (%s rows per GPU)'
                                    % (total_rows, world, '/'.join(str(b - a_) for a_, b in bounds))
                                    if strong else '%d synthetic SPARTA rows/GPU' % rows) +
                                   ' x %d lambda (%.0f-%.0f nm), %d^2 grid, pixscale %.5f, npsflin=%d%s' % (
                                       nl, lb[0], lb[-1], dim, ps, a.npsflin,
                                       ' (BASELINE.json configs[1])'
                                       if (world, rows, nl, dim, a.npsflin) == (1, 100, 35, 512, 1) else
                                       ' (BASELINE.json configs[2])'
                                       if strong and (total_rows, nl, dim, a.npsflin) == (1000, 35, 512, 1) else ''),
                       'rows_total': total_rows, 'rows_per_gpu': rows, 'nl': nl, 'dim': dim, 'npsflin': a.npsflin,
                       'chunk_tasks': chunk, 'parallelism': 'rows sharded x%d' % world},
            # The dominant kernel is the per-wavelength stage: on the matrix
            # cores in mixed mode with one direction, LDS FFTs on the vector pipe otherwise.
            'roofline': {'bound': bound, 'kernel': kernel_name,
                         'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'avg_launch_ms': round(avg_s * 1e3, 4), 'launches': nlaunch, 'launches_timed_every': a.profile_every,
                         'one_call_in_flight': alone_ms and {
                             'avg_launch_ms': round(alone_ms, 4),
                             'achieved': round(flops / (alone_ms * 1e-3) / 1e12, 2),
                             'frac': round(flops / (alone_ms * 1e-3) / 1e12 / peak, 4),
                             'note': 'same launches with nothing else on the GPU (one lane, 40 steps after '
                                     'the timed region); in the timed region the other lane\'s kernels '
                                     'share the CUs'},
                         'nominal_flops_per_launch': flops,
                         'model': model_txt,
                         'fp32_equivalent': flops32 and {
                             'flops_per_launch': flops32,
                             'achieved': round(flops32 / avg_s / 1e12, 2), 'peak': PEAK_FP32_TFLOPS,
                             'frac': round(flops32 / avg_s / 1e12 / PEAK_FP32_TFLOPS, 4),
                             'note': 'one product per element, no padding, against the fp32 matrix / '
                                     'vector peak'},
                         'tile_steps_executed_fraction': mf_work and round(
                             mf_work['tile_steps_per_row'] / mf_work['tile_steps_unpruned_per_row'], 4),
                         'lines_transformed_fraction': round(kept_frac, 4),
                         # what stage A has to deliver at all: blocks some wavelength of the task keeps, and blocks
                         # inside the support of the telescope OTF (fractions of the half plane)
                         'blocks_kept_by_any_wavelength_fraction': round(
                             float(mf_work['union_steps_per_row']) * nl / mf_work['tile_steps_unpruned_per_row'], 4)
                         if (mf_work and mf_work.get('union_steps_per_row', -1) >= 0) else None,
                         'blocks_inside_telescope_support_fraction': round(
                             float(mf_work['support_steps_per_row']) * nl / mf_work['tile_steps_unpruned_per_row'], 4)
                         if (mf_work and mf_work.get('support_steps_per_row', -1) >= 0) else None,
                         'valu_issue': u and u.get('valu_issue'),
                         'mfma_busy': u and u.get('mfma_busy'),
                         'lds_busy': u and u.get('lds_array_busy'),
                         'pmc_source': u and 'profiles/%s (scripts/prof_table.sh, one step in flight)' % util_name,
                         'step': step_util,
                         # the whole step against HBM (the pipe north_star names): algorithmic bytes of the
                         # pipeline as built over the step time; `alone_frac`: the kernel with nothing beside it
                         'hbm_frac': round(model_step / step_s / 1e9 / PEAK_HBM_GBPS, 4),
                         'achieved_GBps': round(model_step / step_s / 1e9, 1),
                         'alone_frac': alone_ms and round(flops / (alone_ms * 1e-3) / 1e12 / peak, 4),
                         'unpruned_fp32_equivalent': (mf_work is not None) and mfma_roofline(
                             mf_work, tasks_per_launch, avg_s * 1e3, dim)['unpruned_fp32_equivalent'],
                         'hbm': {'model_bytes_per_step': model_step, 'model_terms': model,
                                 'traffic_bytes_per_step': traffic_step,
                                 'waste_ratio': traffic_step and round(traffic_step / model_step, 3),
                                 'achieved_GBps': round(model_step / step_s / 1e9, 1), 'peak': PEAK_HBM_GBPS,
                                 'frac': round(model_step / step_s / 1e9 / PEAK_HBM_GBPS, 4),
                                 'note': 'the pipeline moves few bytes by construction (the N^2 row transforms of the PSD '
                                         'are gone with the series form of stage A, the N^2 PSF is never formed): it is '
                                         'bound by the fp64 vector pipe (stage A), the feeding of the matrix cores '
                                         '(per-wavelength stage) and the vector pipe of the fit, not by HBM'}},
            # whole step against HBM: algorithmic bytes of the restructured pipeline, what the
            # PMC counters saw, and the ratio (wasted re-reads / padding / fp64 intermediates)
            'roofline_hbm': {'model_bytes_per_step': model_step, 'model_terms': model,
                             'traffic_bytes_per_step': traffic_step,
                             'waste_ratio': traffic_step and round(traffic_step / model_step, 3),
                             'achieved_GBps': round(model_step / step_s / 1e9, 1),
                             'peak': PEAK_HBM_GBPS,
                             'frac': round(model_step / step_s / 1e9 / PEAK_HBM_GBPS, 4)},
            # SURVEY.md 8(d)'s byte model (full N^2 second pass, which the pruned algorithm never
            # performs): kept only as a labelled algorithmic-savings ratio, not a roofline
            'legacy_model_frac': round((npsf / dt) * legacy_bytes_per_psf / 1e9 / PEAK_HBM_GBPS, 4),
            'kernel_ms_per_step': {k: round(v[0] / max(nprof, 1), 4) for k, v in prof_all.items() if v[1]},
            'kernel_ms_per_step_note': 'second, untimed pass of %d steps with every launch '
                                       'bracketed by HIP events' % nprof,
            # stage A in its series form (this round's kernel): fp64 vector pipe, one call in flight
            'roofline_stage_a': kt_alone and stage_a_roofline(kt_alone, rows * ndir / kl_alone.get('dphi_series', 1), dim, mixed),
            'kernel_ms_one_call_in_flight': {k: round(v, 4) for k, v in kt_alone.items()},
            'fit_iterations': {'mean': round(float(fitg[:, :, 7].mean()), 2),
                               'max': int(fitg[:, :, 7].max())},
            'prime_steps': a.prime,
            'contexts': max(1, a.inflight), 'lanes_per_context': a.streams or 2,
            'host_enqueue_ms_per_step': round(t_enq / a.steps * 1e3, 4),
            'host_enqueue_note': 'host_enqueue_ms_per_step: CPU time of the Python thread per step, '
                                 'which includes spinning while four calls ahead of the GPU; '
                                 'host_library_ms_per_call: wall time inside mpsfr_reconstruct '
                                 'without that wait = what queueing a call costs',
            'host_library_ms_per_call': round(host_lib_s * 1e3, 4),
            'build_id': load_lib().mpsfr_build_id().decode(),
        }
        if rank_dt:
            out['rank_ms_per_step'] = rank_times_block(rank_dt, a.steps)
        if exch_ms:
            out['exchange_ms_per_step'] = round(max(exch_ms), 4)
            out['exchange'] = {'ms_per_step_by_rank': [round(v, 4) for v in exch_ms],
                               'gathered_table_equals_the_last_call': exch_ok,
                               'share_of_step': round(max(exch_ms) / (dt / a.steps * 1e3), 4),
                               'collectives': 'all-gather of the fit tables [rows][nl][16] f64 + sum-reduce of the stamp '
                                              'sums [nl][40][40] f64 to rank 0 (SURVEY.md 8(e))',
                               'how': 'separate untimed pass of %d steps, HIP events on the stream of the collectives '
                                      'around the two calls of every step (they overlap the next call of the library in the '
                                      'timed region: this is their cost, not their share of the critical path)' % min(a.steps, 40)}
        if e2e is not None:
            out['value_e2e_sparta'] = e2e['value']
            out['e2e_sparta'] = e2e
        if len(rep_dt) > 1:
            vals = [npsf / d for d in rep_dt]
            out['timed_region_repeats'] = {'count': len(rep_dt), 'seconds_each': round(dt, 4),
                                           'value_min': round(min(vals), 1), 'value_max': round(max(vals), 1),
                                           'value_median': round(float(np.median(vals)), 1),
                                           'values_first8': [round(v, 1) for v in vals[:8]],
                                           'note': '`value` is the first region; the others repeat the same K steps '
                                                   'behind the same barrier + synchronize brackets'}
        if host_leg is not None:
            out['value_host_outputs'] = host_leg['value']
            out['host_outputs'] = host_leg
        if native is not None:
            out['value_native1280'] = native['value']
            out['native1280'] = native
        if f64 is not None:
            out['value_f64'] = f64['value']
            out['f64'] = f64
        if unpruned is not None:
            out['value_unpruned'] = unpruned['value']
            out['unpruned'] = unpruned
        if cpu is not None:
            out['cpu_baseline'] = cpu
            out['parity'] = parity_block(fitg, cpu_fits.shape[0])
            pm = load_json(PROFILE_ROUND + '_parity_margins.json')
            if pm and 'wide_parameter_range_vs_oracle' in pm.get('tests', {}):
                # worst case of tests/test_gpu_parity.py::test_wide_parameter_range_against_the_oracle
                # (seeing 0.3-2.5", GL 0.02-0.98, L0 8.1-29.9 m), recorded by the test run on the GPU
                out['parity']['wide_parameter_range'] = dict(
                    pm['tests']['wide_parameter_range_vs_oracle'], source='profiles/%s_parity_margins.json' % PROFILE_ROUND)
            out['speedup_vs_cpu_baseline'] = round(out['value'] / cpu['value'], 1)
            # the port is slower than the real reference (profiles/r02_cpu_calibration.json: the reference
            # against the port on this workload's grid, one process): the speed-up against the reference
            ratio = None
            cal = cpu.get('calibration') or {}
            for cfg in cal.get('configs', []):
                if cfg.get('dim') == dim and cfg.get('reference_over_port_time'):
                    ratio = 1.0 / float(cfg['reference_over_port_time'])       # reference PSFs/s over the port's
            if ratio:
                out['speedup_vs_reference_calibrated'] = round(out['value'] / (cpu['value'] * ratio), 1)
                out['speedup_calibration'] = {'reference_psfs_per_s_over_port': round(ratio, 3), 'source': 'profiles/r02_cpu_calibration.json'}
        print(json.dumps(out), flush=True)
    if xchg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
