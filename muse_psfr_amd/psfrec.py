"""Drop-in Python API of the reference's PSF-reconstruction path, on the MI355X.

Mirrors ``muse_psfr.psfrec`` (reference psfrec.py) for this path: ``compute_psf`` (:933-978),
``compute_psf_from_sparta`` (:981-1120), ``create_sparta_table`` (:1123-1141), ``fit_psf_cube``
(:861-871), ``muse_intrinsic_psf`` (:1144-1171), ``fit_psf_with_polynom`` (:1174-1215) -- same names,
argument meaning, return structure, log messages and error behaviour.  All numerics run in
``libmpsfr.so`` (HIP); the reference's joblib fan-out over rows (:1082-1083) becomes one batched
GPU call.  There is no CPU fallback.

Extra keyword-only arguments (defaults = the reference's hard-coded values, so existing callers see
no difference): ``dim=1280`` (:955), ``dimpsf=40`` (:658), ``pixscale=0.2`` (:659, :899, :868),
``precision='mixed'|'f64'``, ``cutoff_masks='host'|'exact'|(rec, res)``, ``device=0``.

astropy is used for tables / FITS when it is importable (then the return types are astropy's, as
in the reference); otherwise the small NumPy implementations in ``_minifits`` / ``Table`` below.
"""
import functools
import logging
import os
from collections import OrderedDict

import numpy as np

from . import _lib, _minifits
from ._lib import Context, MpsfrError, E_GRID

MIN_L0 = 8    # minimum L0 in m (psfrec.py:30)
MAX_L0 = 30   # maximum L0 in m (psfrec.py:31)

logger = logging.getLogger(__name__)

_FIT_COLS = ('lbda', 'center', 'flux', 'fwhm', 'n', 'peak', 'err_center', 'err_flux', 'err_fwhm',
             'err_n', 'err_peak')


_ASTROPY = []


def _astropy():
    """(astropy.io.fits, astropy.table.Table), or (None, None) without a working astropy.  The
    outcome is kept: a failed import costs 0.1 ms every time, a fifth of a one-row compute_psf."""
    if not _ASTROPY:
        try:
            from astropy.io import fits
            from astropy.table import Table as ATable
            _ASTROPY.append((fits, ATable))
        except Exception:  # noqa: BLE001 - absent or broken astropy: use the NumPy implementations
            _ASTROPY.append((None, None))
    return _ASTROPY[0]


class Table:
    """Tiny stand-in for astropy.table.Table (used only when astropy is absent): ordered columns
    of equal length plus a ``meta`` dict."""

    def __init__(self, columns=None, meta=None):
        self.columns = OrderedDict((k, np.asarray(v)) for k, v in (columns or {}).items())
        self.meta = dict(meta or {})

    @property
    def colnames(self):
        return list(self.columns)

    def __len__(self):
        return 0 if not self.columns else len(next(iter(self.columns.values())))

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.columns[key]
        return Table(OrderedDict((k, v[key]) for k, v in self.columns.items()), self.meta)

    def __setitem__(self, key, value):
        n = len(self)
        v = np.asarray(value)
        self.columns[key] = np.full(n, v) if v.ndim == 0 else v

    def __contains__(self, key):
        return key in self.columns

    @classmethod
    def read(cls, hdu):
        d = hdu.data
        meta = {k: v for k, v in hdu.header.items()
                if k not in ('XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'PCOUNT', 'GCOUNT',
                             'TFIELDS', 'EXTNAME') and not k.startswith(('TTYPE', 'TFORM', 'TDIM'))}
        return cls(OrderedDict((n, np.array(d[n])) for n in d.dtype.names), meta)


@functools.lru_cache(maxsize=1)
def host_cutoff_masks():
    """The cut-off masks of psfrec.py:257 (>=) and :435 (>) evaluated with this host's NumPy in
    exactly the reference's way: |f cos(arctan(fy/fx))| and |f sin(arctan(fy/fx))| against
    fc = 1.5.  On the |k| = 24 lines the outcome depends on last-bit libm rounding, so the
    reference itself is platform dependent there ("cut-off masks"); computing them here
    keeps a user's results identical to what the reference gives on the same machine."""
    fx = np.fft.fftfreq(80, 8.0 / 40)[:, np.newaxis]
    fy = fx.T
    f = np.sqrt(fx ** 2 + fy ** 2)
    with np.errstate(all='ignore'):
        arg = fy / fx
    arg[0, 0] = 0
    arg = np.arctan(arg)
    f_x = f * np.cos(arg)
    f_y = f * np.sin(arg)
    fc = 1 / (2 * (8.0 / 24.0))
    rec = (f != 0) & (np.abs(f_x) >= fc) | (np.abs(f_y) >= fc)
    res = (f != 0) & (np.abs(f_x) > fc) | (np.abs(f_y) > fc)
    return rec, res


_contexts = {}


def get_context(dim=1280, pixscale=0.2, dimpsf=40, precision='mixed', device=0, replica=0):
    """Cached GPU context for (dim, pixscale, dimpsf, precision, device); `replica` > 0: a further
    context on the same device (fan-out tests run two shards on one GPU)."""
    key = (int(dim), float(pixscale), int(dimpsf), precision, int(device), int(replica))
    if key not in _contexts:
        _contexts[key] = Context(dim=dim, pixscale=pixscale, dimpsf=dimpsf, precision=precision,
                                 device=device)
    return _contexts[key]


def _resolve_masks(cutoff_masks):
    if cutoff_masks is None or (isinstance(cutoff_masks, str) and cutoff_masks == 'exact'):
        return None
    if isinstance(cutoff_masks, str):
        if cutoff_masks != 'host':
            raise ValueError("cutoff_masks must be 'host', 'exact' or a (rec, res) pair")
        return host_cutoff_masks()
    return cutoff_masks


# Fan-out over several GPUs pays from this many tasks per device (a 100-row batch takes 0.3 ms)
FANOUT_MIN_TASKS_PER_DEVICE = 32


def _fanout_devices(devices, device, ntask, n_jobs):
    """The devices a batch of `ntask` tasks runs on.  An explicit list `devices` is taken as given
    (repeats allowed: several contexts on one GPU).  An explicit `device` is that device and nothing
    else.  Both left at None: every visible GPU when the batch is large enough and the caller did not
    ask for one job (n_jobs = 1; n_jobs > 1 caps the number of devices, like it caps the reference's
    worker processes, psfrec.py:1082) -- unless the process is one rank of a one-rank-per-GPU launch
    (WORLD_SIZE / LOCAL_RANK set, distributed.py): a rank stays on its own device, LOCAL_RANK."""
    if devices is not None:
        devs = [int(d) for d in devices]
        if not devs:
            raise ValueError('devices must not be empty')
        return devs
    if device is not None:
        return [int(device)]
    if 'WORLD_SIZE' in os.environ or 'LOCAL_RANK' in os.environ:
        return [int(os.environ.get('LOCAL_RANK', 0))]
    if n_jobs == 1:
        return [0]
    from ._lib import device_count
    n = device_count()
    if n_jobs is not None and n_jobs > 1:
        n = min(n, int(n_jobs))
    n = min(n, ntask // FANOUT_MIN_TASKS_PER_DEVICE)
    if n <= 1:
        return [0]
    return list(range(n))


def _reconstruct(lbda, tasks, npsflin, h, dim, dimpsf, pixscale, precision, cutoff_masks, device,
                 want_psf=True, devices=None, n_jobs=1):
    if isinstance(tasks, tuple) and len(tasks) == 2 and isinstance(tasks[0], np.ndarray):
        st, t3 = tasks                  # ([ntask][3] seeing / GL / L0, [ntask] three-laser mode)
        see, gl, l0 = (np.ascontiguousarray(st[:, k], dtype=float) for k in range(3))
        three = np.asarray(t3).astype(np.uint8)
    else:                               # list of (seeing, GL, L0, three_lgs_mode)
        see = np.array([t[0] for t in tasks], dtype=float)
        gl = np.array([t[1] for t in tasks], dtype=float)
        l0 = np.array([t[2] for t in tasks], dtype=float)
        three = np.array([1 if t[3] else 0 for t in tasks], dtype=np.uint8)
    masks = _resolve_masks(cutoff_masks)
    devs = _fanout_devices(devices, device, see.size, n_jobs)

    def run(dev, replica, a, b):
        ctx = get_context(dim, pixscale, dimpsf, precision, dev, replica)
        return ctx.reconstruct(lbda, see[a:b], gl[a:b], l0[a:b], three[a:b], h, npsflin=npsflin,
                               masks=masks, want_psf=want_psf)
    try:
        if len(devs) == 1:
            res = run(devs[0], 0, 0, see.size)
            res['devices'] = devs
            return res
        # Row shards over the devices, one context per device and one host thread each inside the
        # library (mpsfr_reconstruct_multi): the reference's joblib fan-out (psfrec.py:1082-1083).
        # Per-task results do not depend on the sharding; the stamp sums are added in device order.
        replica = [devs[:i].count(d) for i, d in enumerate(devs)]
        ctxs = [get_context(dim, pixscale, dimpsf, precision, d, r) for d, r in zip(devs, replica)]
        res = Context.reconstruct_multi(ctxs, lbda, see, gl, l0, three, h, npsflin=npsflin, masks=masks,
                                        want_psf=want_psf)
        res['devices'] = devs
        return res
    except MpsfrError as e:
        if e.code == E_GRID:
            # the reference fails here with a ValueError from scipy's interpn (psfrec.py:663-683)
            raise ValueError(str(e)) from None
        raise


PIPELINE_MIN_TASKS = 125      # a single-device table of at least twice as many tasks goes through asynchronous parts
PIPELINE_PARTS = 2            # one part per pipeline lane (with the FIT_ROWS columns written in C -- mpsfr_fit_rows --
                              # the parts' host work no longer pays for more: 1000 rows x 35 lambda at 512^2 take
                              # 2.41 / 2.49 / 2.62 / 2.64 ms in 2 / 3 / 4 / 5 parts)


def _reconstruct_pipelined(lbda, stats, three, laser_idx, npsflin, h, dim, dimpsf, pixscale, precision,
                           cutoff_masks, dev):
    """compute_psf_from_sparta's batch as asynchronous parts (PIPELINE_PARTS) on one context, the FIT_ROWS records of a
    part assembled while the next parts are on the GPU.  Returns (dict(psf_sum, devices, rec), records)."""
    ntask, nlam = len(stats), lbda.size
    nparts = max(2, min(PIPELINE_PARTS, ntask // PIPELINE_MIN_TASKS))
    bounds = [ntask * k // nparts for k in range(nparts + 1)]
    masks = _resolve_masks(cutoff_masks)
    ctx = get_context(dim, pixscale, dimpsf, precision, dev, 0)
    see, gl, l0 = (np.ascontiguousarray(stats[:, k], dtype=float) for k in range(3))
    t3 = np.asarray(three).astype(np.uint8)
    psum = None
    # Whatever goes wrong between the first asynchronous call and the last wait -- a library error in a
    # later part, a MemoryError in the template, a KeyboardInterrupt -- the parts already queued are
    # abandoned (the library drains and forgets their output arrays, which the context kept alive until
    # now): nothing is left behind that a later call on the cached context could write through.
    try:
        pend = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            pend.append(ctx.reconstruct_async(lbda, see[a:b], gl[a:b], l0[a:b], t3[a:b], h, npsflin=npsflin,
                                              masks=masks, want_psf=False))
        rec, blk = _fit_rows_template(lbda, stats, laser_idx)          # (while the GPU works)
        for (a, b), p in zip(zip(bounds[:-1], bounds[1:]), pend):
            r = p.wait()
            _fit_rows_fill(blk[a * nlam:b * nlam], r['fit'], pixscale)
            psum = r['psf_sum'] if psum is None else psum + r['psf_sum']
    except BaseException as e:
        ctx.abandon()
        if isinstance(e, MpsfrError) and e.code == E_GRID:
            raise ValueError(str(e)) from None
        raise
    return dict(psf_sum=psum, devices=[dev], rec=True), rec


def _fit_columns(lbda, fit, pixscale):
    """fit: (n, NFIT) rows of libmpsfr -> the columns fit_psf_cube keeps (psfrec.py:866-870)."""
    fit = np.asarray(fit)
    n = fit.shape[0]
    cols = OrderedDict()
    cols['lbda'] = np.asarray(lbda, dtype=float)
    cols['center'] = fit[:, 1:3].copy()
    cols['flux'] = fit[:, 15].copy()
    cols['fwhm'] = np.repeat(fit[:, 5:6] * pixscale, 2, axis=1)
    cols['n'] = fit[:, 4].copy()
    cols['peak'] = fit[:, 0].copy()
    cols['err_center'] = fit[:, 9:11].copy()
    with np.errstate(all='ignore'):
        rel = np.sqrt((fit[:, 8] / fit[:, 0]) ** 2 + (2 * fit[:, 11] / fit[:, 3]) ** 2 +
                      (fit[:, 12] / (fit[:, 4] - 1)) ** 2)
    cols['err_flux'] = np.abs(fit[:, 15]) * rel
    cols['err_fwhm'] = np.repeat(fit[:, 13:14] * pixscale, 2, axis=1)
    cols['err_n'] = fit[:, 12].copy()
    cols['err_peak'] = fit[:, 8].copy()
    assert tuple(cols) == _FIT_COLS and all(len(v) == n for v in cols.values())
    return cols


_FIT_ROWS_DTYPE = np.dtype([('lbda', 'f8'), ('center', 'f8', (2,)), ('flux', 'f8'), ('fwhm', 'f8', (2,)),
                            ('n', 'f8'), ('peak', 'f8'), ('err_center', 'f8', (2,)), ('err_flux', 'f8'),
                            ('err_fwhm', 'f8', (2,)), ('err_n', 'f8'), ('err_peak', 'f8'), ('SEEING', 'f8'),
                            ('GL', 'f8'), ('L0', 'f8'), ('row_idx', 'i8'), ('lgs_idx', 'i8')])


def _fit_rows_template(lbda, stats, laser_idx):
    """The FIT_ROWS table of compute_psf_from_sparta (psfrec.py:1086-1101) as one structured array with the
    columns that only depend on the inputs filled in (lbda, SEEING, GL, L0, row_idx, lgs_idx) -- work for
    the time the GPU is busy.  Returns (records, their (n, 20) float64 view)."""
    ntask, nlam = len(stats), len(lbda)
    assert _FIT_ROWS_DTYPE.itemsize == 160 and _FIT_ROWS_DTYPE.names[:11] == _FIT_COLS
    blk = np.empty((ntask * nlam, 20))
    # record layout: 0 lbda, 1-2 center, 3 flux, 4-5 fwhm, 6 n, 7 peak, 8-9 err_center, 10 err_flux,
    # 11-12 err_fwhm, 13 err_n, 14 err_peak, 15 SEEING, 16 GL, 17 L0, 18 row_idx, 19 lgs_idx
    b3 = blk.reshape(ntask, nlam, 20)
    b3[:, :, 0] = np.asarray(lbda, dtype=float)[None, :]
    b3[:, :, 15:18] = np.asarray(stats, dtype=float)[:, None, :]
    ib = blk.view(np.int64).reshape(ntask, nlam, 20)
    ib[:, :, 18] = np.arange(1, ntask + 1)[:, None]
    ib[:, :, 19] = np.asarray(laser_idx, dtype=np.int64)[:, None]
    return blk.view(_FIT_ROWS_DTYPE).reshape(ntask * nlam), blk


def _fit_rows_fill(blk, fit, pixscale):
    """The fit columns of FIT_ROWS (the values of _fit_columns) into the rows `blk` ((n, 20) view of the
    records) from the library's fit rows `fit` (.., NFIT): one pass in C (mpsfr_fit_rows), straight into the
    records -- the NumPy form (a row-wise gather + six strided column operations) took 0.28 ms per 8750 rows,
    a third of the host's critical path behind the last part of a large table."""
    _lib.fit_rows(fit, pixscale, blk[:, 1:15])


def _fit_rows_records(lbda, fit, pixscale, stats, laser_idx):
    """FIT_ROWS from the fit rows of all tasks, (ntask, nl, NFIT)."""
    rec, blk = _fit_rows_template(lbda, stats, laser_idx)
    _fit_rows_fill(blk, fit, pixscale)
    return rec


def _make_table(cols, meta=None):
    _, ATable = _astropy()
    if ATable is not None:
        t = ATable(cols)
        t.meta.update(meta or {})
        return t
    return Table(cols, meta)


def fit_psf_cube(lbda, psfcube, *, pixscale=0.2, precision='mixed', device=0):
    """Fit a Moffat PSF on each wavelength plane of the psfcube (psfrec.py:861-871)."""
    data = np.asarray(getattr(psfcube, 'data', psfcube), dtype=float)
    ctx = get_context(128, pixscale, data.shape[-1], precision, device)
    return _make_table(_fit_columns(lbda, ctx.fit_stamps(data), pixscale))


def simul_psd_wfm(Cn2, h, seeing, L0, zenith=0., plot=False, npsflin=1, dim=1280, three_lgs_mode=False,
                  verbose=True, *, precision='mixed', cutoff_masks='host', device=0):
    """Residual phase PSD of the MUSE wide-field mode for each evaluation direction (psfrec.py:36-151):
    (npsflin**2, dim, dim) float64, DC at [dim/2, dim/2], in nm^2 m^2 like the reference's.  Two layers
    (`Cn2` = their weights, normalised here as psfrec.py:57-58 does); the zenith angle only rescales r0
    (psfrec.py:108, 183-187)."""
    Cn2 = np.array(Cn2, dtype=float)
    if Cn2.size != 2 or len(h) != 2:
        raise ValueError('exactly two layers are supported (psfrec.py:66 fixes two wind directions)')
    Cn2 = Cn2 / Cn2.sum()
    if verbose and three_lgs_mode:
        logger.info('Using three lasers mode')
    if plot:
        direction_perf(npsflin, plot=True)
    seeing_los = float(seeing) / np.cos(np.deg2rad(zenith)) ** (3 / 5)
    ctx = get_context(dim, 0.2, 40, precision, device)
    return ctx.simul_psd(seeing_los, Cn2[0], L0, three_lgs_mode, h, npsflin=npsflin,
                         masks=_resolve_masks(cutoff_masks))


def psf_muse(psd, lambdamuse, *, pixscale=0.2, precision='mixed', device=0):
    """PSF stamps (nl, 40, 40) at the wavelengths `lambdamuse` [nm] from a residual PSD (psfrec.py:644-686:
    structure function, OTF, crop to the field of the stamp, bilinear sampling; the mean over the
    directions when the PSD has three dimensions).  Any real PSD image of a supported size."""
    psd = np.asarray(psd, dtype=float)
    ctx = get_context(psd.shape[-1], pixscale, 40, precision, device)
    try:
        return ctx.psf_from_psd(psd, np.atleast_1d(np.asarray(lambdamuse, dtype=float)))
    except MpsfrError as e:
        if e.code == E_GRID:
            raise ValueError(str(e)) from None
        raise


def convolve_final_psf(lbda, seeing, GL, L0, psf, *, pixscale=0.2, precision='mixed', device=0):
    """Convolve with the tip-tilt and MUSE PSFs to get the final PSF (psfrec.py:874-930).  `psf`:
    (nl, 40, 40)."""
    psf = np.asarray(psf, dtype=float)
    ctx = get_context(128, pixscale, psf.shape[-1], precision, device)
    return ctx.convolve_stamps(lbda, seeing, GL, L0, psf)


def compute_psf(lbda, seeing, GL, L0, npsflin=1, h=(100, 10000), three_lgs_mode=False,
                verbose=True, *, dim=1280, dimpsf=40, pixscale=0.2, precision='mixed',
                cutoff_masks='host', device=0):
    """Reconstruct a PSF from a set of seeing, GL, and L0 values (psfrec.py:933-978).

    Returns ``(table, psf)``: the per-wavelength Moffat fit table (with SEEING, GL, L0 columns and
    meta) and the (nl, 40, 40) float64 PSF cube."""
    lbda = np.atleast_1d(np.asarray(lbda, dtype=float))
    if verbose:
        logger.info('Compute PSF with seeing=%.2f GL=%.2f L0=%.2f', seeing, GL, L0)
        if three_lgs_mode:
            logger.info('Using three lasers mode')
    r = _reconstruct(lbda, [(seeing, GL, L0, three_lgs_mode)], npsflin, h, dim, dimpsf, pixscale,
                     precision, cutoff_masks, device)
    cols = _fit_columns(lbda, r['fit'][0], pixscale)
    nl = lbda.size
    cols['SEEING'] = np.full(nl, float(seeing))
    cols['GL'] = np.full(nl, float(GL))
    cols['L0'] = np.full(nl, float(L0))
    res = _make_table(cols, {'SEEING': float(seeing), 'GL': float(GL), 'L0': float(L0)})
    return res, r['psf'][0]


def _table_hdu(cols, meta, name):
    fits, ATable = _astropy()
    if fits is not None:
        t = ATable(cols)
        t.meta.update(meta)
        hdu = fits.table_to_hdu(t)
        hdu.name = name
        return hdu
    hdr = _minifits.Header()
    for k, v in meta.items():
        hdr[k] = v
    return _minifits.BinTableHDU.from_columns(cols, hdr, name)


def compute_psf_from_sparta(filename, extname='SPARTA_ATM_DATA', npsflin=1, lmin=490, lmax=930,
                            nl=35, lbda=None, h=(100, 10000), n_jobs=-1, plot=False,
                            mean_of_lgs=True, verbose=True, *, dim=1280, dimpsf=40, pixscale=0.2,
                            precision='mixed', cutoff_masks='host', device=None, devices=None):
    """Reconstruct a PSF from SPARTA data (psfrec.py:981-1120).

    ``filename`` is a FITS path or an already opened HDUList.  Returns an HDUList with
    PRIMARY, a copy of the SPARTA extension, FIT_ROWS, FIT_MEAN and PSF_MEAN -- or ``None``
    (with a 'No valid values' warning) when no row has a valid laser.  The rows are processed as one
    GPU batch -- or, like the reference's ``n_jobs`` worker processes (psfrec.py:1082-1083), as one
    batch per GPU: with ``device`` and ``devices`` both None every visible GPU is taken when the table
    has at least FANOUT_MIN_TASKS_PER_DEVICE tasks per device (``n_jobs`` = 1 keeps one device,
    ``n_jobs`` > 1 caps their number; a rank of a one-process-per-GPU launch keeps its own);
    ``device=k`` names the one device to use, ``devices=[...]`` several.  The per-row results do not
    depend on the split."""
    fits, _ = _astropy()
    io_mod = fits if fits is not None else _minifits
    opened = False
    if isinstance(filename, (list, _minifits.HDUList)) or (
            fits is not None and isinstance(filename, fits.HDUList)):
        hdul = filename
    else:
        hdul = io_mod.open(filename)
        opened = True
    try:
        ext = hdul[extname]
        data = np.array(ext.data)
        if fits is not None and isinstance(ext, fits.BinTableHDU):
            out = fits.HDUList([fits.PrimaryHDU(), ext.copy()])
        else:
            out = _minifits.HDUList([_minifits.PrimaryHDU(), ext.copy()])
    finally:
        if opened:
            hdul.close()

    nrows = len(data)
    if nrows == 1:
        n_jobs = 1
    if lbda is None:
        lbda = np.linspace(lmin, lmax, nl)
    lbda = np.atleast_1d(np.asarray(lbda, dtype=float))
    if verbose:
        logger.info('Processing SPARTA table with %d values, njobs=%d ...', nrows, n_jobs)

    # The rows are walked by NumPy, not by a Python loop (a 1000-row table spent as long in that loop
    # as on the GPU): values[row][laser] = (seeing, GL, L0), the outlier rejection of psfrec.py:1049-1051,
    # and the tasks in the reference's order -- rows ascending, lasers ascending inside a row.
    values = np.empty((nrows, 4, 3), dtype=float)
    for k in range(4):
        for j, col in enumerate(('SEEING', 'TUR_GND', 'L0')):
            values[:, k, j] = data['LGS%d_%s' % (k + 1, col)]
    ok = (values[:, :, 1] > 0) & (values[:, :, 2] < MAX_L0) & (values[:, :, 2] > MIN_L0)
    nb_gs = ok.sum(axis=1)
    if verbose:
        for irow in np.nonzero(nb_gs < 4)[0]:
            if nb_gs[irow] == 0:
                logger.info('%d/%d : No valid values, skipping this row', irow + 1, nrows)
            else:
                logger.info('%d/%d : Using only %d values out of 4 after outliers rejection',
                            irow + 1, nrows, nb_gs[irow])
    if mean_of_lgs:
        rows = np.nonzero(nb_gs > 0)[0]
        # the mean over the valid lasers, added in laser order like values[ok].mean(axis=0)
        stats = np.where(ok[rows][:, :, None], values[rows], 0.0).sum(axis=1) / nb_gs[rows][:, None]
        laser_idx = np.full(rows.size, -1)
    else:
        rows, las = np.nonzero(ok)
        stats = values[rows, las]
        laser_idx = las + 1
    three = nb_gs[rows] < 4
    to_compute = (stats, three)

    if len(stats) == 0:
        logger.warning('No valid values')
        return None

    if verbose:
        for (seeing, GL, L0), t3 in zip(stats, three):
            logger.info('Compute PSF with seeing=%.2f GL=%.2f L0=%.2f', seeing, GL, L0)
            if t3:
                logger.info('Using three lasers mode')

    ntask, nlam = len(stats), lbda.size
    devs = _fanout_devices(devices, device, ntask, n_jobs)
    if len(devs) == 1 and ntask >= 2 * PIPELINE_MIN_TASKS:
        # A large table on one device goes through the library as up to four asynchronous host-output
        # calls (mpsfr_reconstruct on_device = 2): the records of one part are assembled while the GPU
        # works on the next ones.  Per-task results do not depend on the split; the stamp sums of the
        # parts are added in order.
        r, rec = _reconstruct_pipelined(lbda, stats, three, laser_idx, npsflin, h, dim, dimpsf, pixscale, precision,
                                        cutoff_masks, devs[0])
        # (with astropy -- the reference's environment -- the same record array becomes an astropy table HDU)
        out.append(_minifits.BinTableHDU(rec, _minifits.Header(), 'FIT_ROWS') if fits is None
                   else fits.BinTableHDU(data=rec, name='FIT_ROWS'))
    else:
        r = _reconstruct(lbda, to_compute, npsflin, h, dim, dimpsf, pixscale, precision, cutoff_masks,
                         device, want_psf=False, devices=devices, n_jobs=n_jobs)

    # FIT_ROWS: the per-task tables stacked (psfrec.py:1086-1101)
    if 'rec' in r:
        pass
    elif fits is None:
        out.append(_minifits.BinTableHDU(_fit_rows_records(lbda, r['fit'], pixscale, stats, laser_idx),
                                         _minifits.Header(), 'FIT_ROWS'))
    else:
        cols = _fit_columns(np.tile(lbda, ntask), r['fit'].reshape(ntask * nlam, -1), pixscale)
        cols['SEEING'] = np.repeat(stats[:, 0], nlam)
        cols['GL'] = np.repeat(stats[:, 1], nlam)
        cols['L0'] = np.repeat(stats[:, 2], nlam)
        cols['row_idx'] = np.repeat(np.arange(1, ntask + 1), nlam)
        cols['lgs_idx'] = np.repeat(np.asarray(laser_idx), nlam)
        out.append(_table_hdu(cols, {}, 'FIT_ROWS'))

    # mean PSF over the tasks and its fit (psfrec.py:1104-1113)
    psftot = r['psf_sum'] / ntask
    ctx = get_context(dim, pixscale, dimpsf, precision, r.get('devices', [device or 0])[0])
    mcols = _fit_columns(lbda, ctx.fit_stamps(psftot), pixscale)
    seeing, GL, L0 = np.median(stats, axis=0)
    out.append(_table_hdu(mcols, {'SEEING': float(seeing), 'GL': float(GL), 'L0': float(L0)},
                          'FIT_MEAN'))
    if fits is not None and isinstance(out, fits.HDUList):
        out.append(fits.ImageHDU(data=psftot, name='PSF_MEAN'))
    else:
        out.append(_minifits.ImageHDU(data=psftot, name='PSF_MEAN'))

    if plot:
        import matplotlib.pyplot as plt
        plot_psf(out, npsflin=npsflin)
        plt.show()
    return out


def create_sparta_table(nlines=1, seeing=1, L0=25, GL=0.7, bad_l0=False, outfile=None):
    """Helper to create a SPARTA table with the given seeing, L0, and GL values for the 4 LGS
    (psfrec.py:1123-1141).  Returns the table HDU named SPARTA_ATM_DATA."""
    cols = OrderedDict()
    for k in range(1, 5):
        for col, v in (('SEEING', seeing), ('TUR_GND', GL), ('L0', L0)):
            cols['LGS%d_%s' % (k, col)] = np.full(nlines, float(v))
    if bad_l0:
        cols['LGS4_L0'] = np.full(nlines, 150.0)
    hdu = _table_hdu(cols, {}, 'SPARTA_ATM_DATA')
    if outfile is not None:
        fits, _ = _astropy()
        if fits is not None:
            hdu.writeto(outfile, overwrite=True)
        else:
            _minifits.HDUList([_minifits.PrimaryHDU(), hdu]).writeto(outfile, overwrite=True)
    return hdu


def muse_intrinsic_psf(lbda):
    """MUSE PSF polynomial approximation (psfrec.py:1144-1171): fwhm, beta, fwhm_std, beta_std."""
    pol_beta = [-0.83704697, 1.1337153, 0.0609222, -1.35581762, 1.15237178, 2.2106042]
    pol_fwhm = [0.60467385, -1.58905792, 1.75293264, -1.0368302, 0.21487023, 0.34851139]
    pol_beta_std = [0.18187424, -0.17841793, 0.30962616]
    pol_fwhm_std = [0.00707504, -0.0303464, 0.04596354]
    lb = (10 * np.asarray(lbda, dtype=float) - 4750) / (9350 - 4750)
    return (np.polyval(pol_fwhm, lb), np.polyval(pol_beta, lb), np.polyval(pol_fwhm_std, lb),
            np.polyval(pol_beta_std, lb))


def fit_psf_with_polynom(lbda, fwhm, beta, deg=(5, 5), output=0):
    """Fit MUSE PSF fwhm and beta with polynoms (psfrec.py:1174-1215)."""
    def norm(x):
        return (np.asarray(x, dtype=float) - 475) / (935 - 475) - 0.5
    fwhm_pol = np.polyfit(norm(lbda), fwhm, deg[0])
    beta_pol = np.polyfit(norm(lbda), beta, deg[1])
    res = dict(fwhm_pol=fwhm_pol, beta_pol=beta_pol, lbda=lbda, lbda_lim=(475, 935))
    if output > 0:
        lbda_fit = np.linspace(475, 935, 50)
        res['lbda_fit'] = lbda_fit
        res['fwhm_fit'] = np.polyval(fwhm_pol, norm(lbda_fit))
        res['beta_fit'] = np.polyval(beta_pol, norm(lbda_fit))
    return res


def direction_perf(npts, field_size=60, plot=False, lgs=None, ngs=None, ax=None):
    """Grid of directions (arcsec) where the PSF is estimated (psfrec.py:154-180): (2, npts^2),
    (mgrid - npts // 2) * field_size / 2.  With ``plot`` the directions (and the guide stars, if
    given) are drawn on ``ax``."""
    gx, gy = (np.mgrid[:npts, :npts] - npts // 2) * field_size / 2
    dirperf = np.array([gx, gy]).reshape(2, -1)
    if plot:
        import matplotlib.pyplot as plt
        if ax is None:
            _, ax = plt.subplots()
        extent = np.max(dirperf)
        ax.scatter(dirperf[0], dirperf[1], marker='o', s=10, label='Reconstruction directions')
        for stars, size, label in ((lgs, 60, 'LGS'), (ngs, 40, 'NGS')):
            if stars is not None:
                extent = max(extent, np.max(stars))
                ax.scatter(stars[0], stars[1], marker='*', s=size, label=label)
        ax.set_xlim((-1.25 * extent, 1.25 * extent))
        ax.set_ylim((-1.25 * extent, 1.25 * extent))
        ax.set_xlabel('arcsecond')
        ax.set_ylabel('arcsecond')
        ax.legend(loc='upper center')
    return dirperf


def radial_profile(arr, binsize=1):
    """Azimuthal mean of ``arr`` in rings of width ``binsize`` around pixel
    (int(n0/2 + .5), int(n1/2 + .5)) (psfrec.py:810-823).  Returns (bin centres, mean per ring);
    empty rings give nan, as in the reference."""
    arr = np.asarray(arr, dtype=float)
    c0, c1 = int(arr.shape[0] / 2 + .5), int(arr.shape[1] / 2 + .5)
    r = np.hypot(np.arange(arr.shape[0])[:, None] - c0, np.arange(arr.shape[1])[None, :] - c1)
    nbins = int(np.round(r.max() / binsize) + 1)
    edges = np.linspace(0, nbins * binsize, nbins + 1)
    count = np.histogram(r, edges)[0]
    total = np.histogram(r, edges, weights=arr)[0]
    with np.errstate(invalid='ignore', divide='ignore'):
        return (edges[1:] + edges[:-1]) / 2, total / count


def plot_psf(filename, npsflin=1):
    """Figure of a reconstruction result (psfrec.py:826-858): 2 x 3 panels -- the second plane of
    PSF_MEAN (log scale), an empty panel, the reconstruction directions with the four LGS at 63
    arcsec; the radial profile of that plane (log), FWHM(lambda) and beta(lambda) from FIT_MEAN.
    ``filename`` is an HDUList (astropy's or this package's) or a path to the FITS file."""
    import matplotlib.pyplot as plt
    from matplotlib.colors import LogNorm
    opened = isinstance(filename, (str, os.PathLike))
    hdul = (_astropy()[0] or _minifits).open(filename) if opened else filename
    try:
        psf = np.array(hdul['PSF_MEAN'].data, dtype=float)
        fit = hdul['FIT_MEAN'].data
        lbda = np.array(fit['lbda'], dtype=float)
        fwhm = np.array(fit['fwhm'], dtype=float)[:, 0]
        beta = np.array(fit['n'], dtype=float)
    finally:
        if opened:
            hdul.close()
    plane = psf[1]
    fig, axes = plt.subplots(2, 3, figsize=(12, 6), tight_layout=True)
    top, bottom = axes
    im = top[0].imshow(plane, origin='lower', norm=LogNorm())
    fig.colorbar(im, ax=top[0])
    top[0].set_title('PSF')
    top[1].axis('off')
    poslgs = 63.0 * np.array([[1, 1], [-1, -1], [-1, 1], [1, -1]], dtype=float).T   # arcsec
    direction_perf(npsflin, plot=True, lgs=poslgs, ax=top[2])
    centers, prof = radial_profile(plane)
    bottom[0].plot(centers[1:], prof[1:], lw=1)
    bottom[0].set_yscale('log')
    bottom[0].set_title('radial profile')
    bottom[1].plot(lbda, fwhm)
    bottom[1].set_title(r'$FWHM(\lambda)$')
    bottom[2].plot(lbda, beta)
    bottom[2].set_title(r'$\beta(\lambda)$')
    return fig
