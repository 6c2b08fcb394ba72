#!/usr/bin/env python3
"""Benchmark of the PSF-reconstruction hot path (BASELINE.json metric: PSFs/sec, one PSF = one
(SPARTA row x wavelength) final 40x40 stamp including convolutions and Moffat fit).

    python bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling): BASELINE.json configs[1] per GPU -- 100 synthetic SPARTA
rows x 35 wavelengths (465-930 nm) on a 512^2 grid, pixscale 0.2*512/1344,
npsflin=1.  One step = one pass of the hot path over the rank's 100-row batch plus, for N > 1,
the RCCL all-gather of the fit tables and sum-reduce of the partial mean-PSF numerators.
Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the roofline definition.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
DOMINANT = 'otf_rowfft'      # the kernel the roofline object describes
PRIME_STEPS = 64             # untimed, before the warm-up steps
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
    core over rows like the reference's joblib fan-out, psfrec.py:1082-1083) on `nrows` rows."""
    import multiprocessing as mp
    jobs = [(lb, see[i], gl[i], l0[i], dim, ps, True) for i in range(nrows)]
    ctx = mp.get_context('fork')
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_rows, jobs[:min(cores, len(jobs))][:1])      # warm imports
        t = time.time()
        res = pool.map(_cpu_rows, jobs, chunksize=1)
        dt = time.time() - t
    fits = np.array([r[0] for r in res])
    return dict(value=nrows * lb.size / dt, seconds=dt), fits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rows', type=int, default=100, help='rows per GPU per step')
    ap.add_argument('--dim', type=int, default=512)
    ap.add_argument('--nl', type=int, default=35)
    ap.add_argument('--npsflin', type=int, default=1)
    ap.add_argument('--precision', default='mixed', choices=['mixed', 'f64'])
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--fast-exp', type=int, default=1)
    ap.add_argument('--streams', type=int, default=0, help='pipeline lanes (0: library default)')
    ap.add_argument('--inflight', type=int, default=2,
                    help='steps in flight: contexts (stream + workspaces) fed in turn')
    ap.add_argument('--cpu-rows', type=int, default=-1,
                    help='rows of the CPU-baseline sample (-1: two per core, 0: skip)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run' % a.gpus)
        a.gpus = world

    from muse_psfr_amd import synthetic_rows, grid_pixscale
    dim, nl, rows = a.dim, a.nl, a.rows
    ps = grid_pixscale(dim)
    lb = np.linspace(465.0, 930.0, nl) if dim != 1280 else np.linspace(490.0, 930.0, nl)
    see, gl, l0 = synthetic_rows(rows * world)
    sl = slice(rank * rows, (rank + 1) * rows)

    # ---- CPU baseline first (fork pool, before this process touches the GPU)
    cpu = None
    cpu_fits = None
    if rank == 0 and world == 1 and a.cpu_rows != 0 and a.npsflin == 1:
        # a 1-GPU box has a 16-core CPU share whatever os.cpu_count() says; override with
        # MPSFR_BENCH_CORES
        cores = int(os.environ.get('MPSFR_BENCH_CORES', min(os.cpu_count() or 1, 16)))
        ncpu = 2 * cores if a.cpu_rows < 0 else a.cpu_rows
        ncpu = min(ncpu, rows)
        r, cpu_fits = cpu_baseline(lb, see, gl, l0, dim, ps, ncpu, cores)
        cpu = dict(value=round(r['value'], 3), unit='PSFs/sec', cores=cores, kind='port',
                   sample='%d rows x %d lambda on %d^2 (first rows of the GPU workload), '
                          'oracle/psfr_oracle.py reference-shaped (4 FFTs/lambda, fp64, scipy '
                          'leastsq fit), one process per core, %.1f s wall' % (
                              ncpu, nl, dim, r['seconds']))

    import torch
    import torch.distributed as dist
    from muse_psfr_amd import Context, NFIT
    # one rank per GPU; the modulo only matters when rehearsing N > 1 on a box with fewer GPUs
    # (MPSFR_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device)
    local = local % torch.cuda.device_count()
    backend = os.environ.get('MPSFR_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    # Steps are independent batches, so they are pipelined through `--inflight` contexts (each
    # with its own HIP stream and workspaces) fed in turn: while one step is in its fit the next
    # is in its transforms, and LDS-, VALU- and HBM-bound kernels of different steps share the GPU
    # (two contexts: +24 % over one [measured]).  A step is still one mpsfr_reconstruct of the
    # rank's rows; every step's outputs are produced.
    NCTX = max(1, a.inflight)
    ctxs = []
    for _ in range(NCTX):
        c = Context(dim=dim, pixscale=ps, precision=a.precision, device=local)
        if a.chunk:
            c.set_option('chunk_tasks', a.chunk)
        c.set_option('fast_exp', a.fast_exp)
        if a.streams:
            c.set_option('streams', a.streams)
        ctxs.append(c)
    ctx = ctxs[0]

    from muse_psfr_amd.distributed import gather_fit_tables, reduce_psf_sum
    # One set of result buffers per context; with N > 1 the exchange of step i (on torch's stream)
    # overlaps the reconstruction of the following steps (on the other contexts' streams).
    NBUF = NCTX
    fits = [torch.zeros((rows, nl, NFIT), dtype=torch.float64, device=dev) for _ in range(NBUF)]
    psums = [torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev) for _ in range(NBUF)]
    fit = fits[0]
    three = np.zeros(rows, np.uint8)
    h = (100, 10000)
    state = {'i': 0, 'ev': [None] * NBUF}

    # Each context runs on its own HIP stream; torch orders its collectives against it on the GPU
    # (no host sync inside a step): torch's stream waits for the context's stream before the
    # exchange, and that stream waits for the exchange that last read its buffer set before the
    # set is overwritten.
    lib_streams = [torch.cuda.ExternalStream(c.stream_handle(), device=dev) for c in ctxs]

    def step():
        b = state['i'] % NBUF
        state['i'] += 1
        fit_b, psum_b = fits[b], psums[b]
        if state['ev'][b] is not None:
            lib_streams[b].wait_event(state['ev'][b])
        ctxs[b].reconstruct_device(lb, see[sl], gl[sl], l0[sl], three, h, 12.0, a.npsflin, None,
                                   None, psum_b.data_ptr(), fit_b.data_ptr())
        if world > 1:      # FIT_ROWS gather + PSF_MEAN numerator reduce
            cur = torch.cuda.current_stream()
            cur.wait_stream(lib_streams[b])
            if backend == 'nccl':
                state['fit_all'] = gather_fit_tables(fit_b, world * rows)
                reduce_psf_sum(psum_b, dst=0)
            else:          # CPU rehearsal of the same exchange
                state['fit_all'] = gather_fit_tables(fit_b.cpu(), world * rows)
                state['psum'] = reduce_psf_sum(psum_b.cpu(), dst=0)
            ev = torch.cuda.Event()
            ev.record(cur)
            state['ev'][b] = ev

    def fence():
        torch.cuda.synchronize()
        for c in ctxs:
            c.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the steps are queued from Python: a cyclic-GC pass in the middle of the timed loop (tens of
    # ms with torch loaded) would starve the GPU, so collection is parked for the measurement
    gc.collect()
    gc.disable()
    # untimed priming before the W warm-up steps: lets the HIP runtime grow its command/signal
    # pools to the depth the host runs ahead by, and the GPU leave its idle clocks after the CPU
    # baseline (the first ~50 calls of a process are 5-10 % slower)
    for _ in range(PRIME_STEPS):
        step()
    fence()
    for _ in range(a.warmup):
        step()
    # Timed region: HIP events only around the dominant kernel (roofline.achieved); bracketing
    # every launch costs ~8 % of a step in event packets, so the per-kernel table comes from a
    # second, untimed pass of the same K steps.
    def profile_sum():
        tot = {}
        for c in ctxs:
            for k, (ms, n) in c.profile().items():
                t = tot.get(k, (0.0, 0))
                tot[k] = (t[0] + ms, t[1] + n)
        return tot

    for c in ctxs:
        c.set_option('profile_only', c.profile_names().index(DOMINANT))
        c.set_option('profile', 1)
        c.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enq = time.perf_counter() - t0      # host time to queue the K steps (no GPU wait inside)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    prof = profile_sum()
    for c in ctxs:
        c.set_option('profile_only', -1)
        c.profile_reset()
    for _ in range(a.steps):
        step()
    fence()
    prof_all = profile_sum()
    for c in ctxs:
        c.set_option('profile', 0)
    gc.enable()

    if rank == 0:
        npsf = world * rows * nl * a.steps
        ndir = a.npsflin ** 2
        p = 4 if a.precision == 'mixed' else 8
        # dominant kernel: otf_rowfft.  Algorithmic bytes per (task, dir, lambda) for this kernel:
        # read D_phi0 (p N^2) + write the half-plane intermediate (p N^2) = 2 p N^2 of the
        # 3 p N^2 of SURVEY.md 8(d); the third p N^2 (reading it back) belongs to colpass.
        ms, nlaunch = prof[DOMINANT]
        chunk = a.chunk or 'auto'
        units_per_launch = rows * nl * ndir * a.steps / max(nlaunch, 1)
        alg_bytes = 2 * p * dim * dim * units_per_launch
        avg_s = ms / max(nlaunch, 1) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        bytes_per_psf = ndir * dim * dim * (3 * p + (5 * 8 + p) / nl)
        pipe = (npsf / dt) * bytes_per_psf / 1e9
        # what actually bounds the dominant kernel (PMC passes committed under profiles/)
        measured_bound = None
        ufile = os.path.join(ROOT, 'profiles', 'r01_kernel_util.json')
        if os.path.exists(ufile) and dim == 512 and a.precision == 'mixed':
            u = json.load(open(ufile)).get('k_' + DOMINANT)
            if u:
                measured_bound = {'kernel': DOMINANT, 'lds_array_busy': u['lds_array_busy'],
                                  'valu_issue': u['valu_issue'],
                                  'lds_conflict_share': u['lds_conflict_share'],
                                  'hbm_GBps': u['hbm_GBps'],
                                  'source': 'profiles/r01_kernel_util.json (scripts/prof_table.sh)'}
        # HBM traffic of the dominant kernel from the committed PMC pass (profiles/), if it was
        # taken on this workload: (FETCH_SIZE + WRITE_SIZE) KiB per launch, no width correction
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
        if os.path.exists(tfile) and (dim, nl, rows, a.npsflin, a.precision) == (512, 35, 100, 1, 'mixed'):
            tj = json.load(open(tfile))
            k = tj['kernels'].get('k_otf_rowfft', {})
            if k.get('fetch_kib') and k.get('write_kib'):
                # per-launch counters of the profiled run, rescaled to this run's launch size
                per_unit = (k['fetch_kib'] + k['write_kib']) * 1024.0 / tj['units_per_launch']
                traffic = per_unit * (rows * nl * ndir * a.steps / max(prof['otf_rowfft'][1], 1))
        fitg = fit.cpu().numpy()
        out = {
            'metric': 'PSFs/sec (row x lambda) on %d^2 grid, %d lambda' % (dim, nl),
            'value': round(npsf / dt, 1), 'unit': 'PSFs/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 PSD->structure function, f32 per-lambda OTF/FFT, f64 fit'
                     if a.precision == 'mixed' else 'f64',
            'data': 'synthetic',
            'config': {'workload': '%d synthetic SPARTA rows/GPU x %d lambda (%.0f-%.0f nm), '
                                   '%d^2 grid, pixscale %.5f, npsflin=%d (BASELINE.json '
                                   'configs[1])' % (rows, nl, lb[0], lb[-1], dim, ps, a.npsflin),
                       'rows_per_gpu': rows, 'nl': nl, 'dim': dim, 'npsflin': a.npsflin,
                       'chunk_tasks': chunk, 'parallelism': 'rows sharded x%d' % world},
            'roofline': {'bound': 'hbm', 'kernel': DOMINANT,
                         'achieved': round(achieved, 1), 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': round(achieved / 8000.0, 4), 'traffic': traffic,
                         'avg_launch_ms': round(avg_s * 1e3, 4), 'launches': nlaunch,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'note': 'algorithmic bytes = SURVEY 8(d) figure for this kernel (2 p N^2 per '
                                 'task x dir x lambda, i.e. full-plane passes); the restructured kernel '
                                 'never materialises them (traffic = measured FETCH_SIZE + WRITE_SIZE), '
                                 'so frac > 1; it is LDS-store/VALU bound (profiles/r01_pmc_summary.txt, '
                                 'DESIGN.md section 5)'},
            'measured_bound': measured_bound,
            'roofline_pipeline': {'bytes_per_psf': bytes_per_psf,
                                  'achieved_GBps': round(pipe, 1),
                                  'frac_of_8TBps': round(pipe / 8000.0, 4),
                                  'frac_of_6.29TBps': round(pipe / 6290.0, 4)},
            'kernel_ms_per_step': {k: round(v[0] / a.steps, 4) for k, v in prof_all.items() if v[1]},
            'prime_steps': PRIME_STEPS,
            'inflight_steps': NCTX,
            'host_enqueue_ms_per_step': round(t_enq / a.steps * 1e3, 4),
            'kernel_ms_per_step_note': 'second, untimed pass of the same steps with every launch '
                                       'bracketed by HIP events',
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
            n = cpu_fits.shape[0]
            out['parity'] = {
                'rows_checked': n,
                'max_abs_err_fwhm_arcsec': float(np.abs(fitg[:n, :, 5] * ps - cpu_fits[:, :, 3]).max()),
                'max_abs_err_beta': float(np.abs(fitg[:n, :, 4] - cpu_fits[:, :, 4]).max()),
                'tolerance': 1e-4}
            out['speedup_vs_cpu_baseline'] = round(out['value'] / cpu['value'], 1)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
