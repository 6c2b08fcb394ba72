"""Driver of tests/test_astropy_env.py: runs under an interpreter that HAS astropy (the reference's own
environment: /opt/conda/bin/python3.9 in the build container) and exercises the astropy branches of the drop-in
layer (muse_psfr_amd/psfrec.py: _astropy, _make_table, _table_hdu, the fits.BinTableHDU(data=rec) branch of the
asynchronous large-table path, fits.HDUList / ImageHDU) WITHOUT a GPU: the GPU context is replaced by a stub that
returns canned fit rows and stamp sums built from the reference's own outputs (tests/golden/g7_sparta_lgs.npz),
so what comes out of compute_psf_from_sparta can be compared with what the reference returned.

Prints one JSON document on stdout.  usage: python astropy_env_driver.py <repo root>"""
import json
import os
import sys

import numpy as np

# astropy 4.3.1 (the build container's) references names NumPy >= 1.23 removed
if not hasattr(np, 'asscalar'):
    np.asscalar = lambda a: a.item()
if not hasattr(np, 'alen'):
    np.alen = len

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)

from astropy.io import fits                      # noqa: E402
from astropy.table import Table as ATable        # noqa: E402

import muse_psfr_amd.psfrec as P                 # noqa: E402
from muse_psfr_amd._lib import NFIT              # noqa: E402

G7 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g7_sparta_lgs.npz'))
PIX = 0.2


def fit_rows_from_columns(tag, n):
    """Library fit rows (n, NFIT) whose FIT_ROWS / FIT_MEAN columns are the reference's (psfrec.py:866-870):
    the inverse of psfrec._fit_columns / mpsfr_fit_rows."""
    g = lambda c: np.asarray(G7['%s_%s' % (tag, c)])[:n]      # noqa: E731
    f = np.zeros((n, NFIT))
    f[:, 0] = g('peak')
    f[:, 1:3] = g('center')
    f[:, 3] = 1.0                                             # a (only err_a / a enters err_flux)
    f[:, 4] = g('n')
    f[:, 5] = g('fwhm')[:, 0] / PIX
    f[:, 8] = g('err_peak')
    f[:, 9:11] = g('err_center')
    f[:, 12] = g('err_n')
    f[:, 13] = g('err_fwhm')[:, 0] / PIX
    f[:, 15] = g('flux')
    rel2 = (g('err_flux') / g('flux')) ** 2 - (f[:, 8] / f[:, 0]) ** 2 - (f[:, 12] / (f[:, 4] - 1)) ** 2
    f[:, 11] = 0.5 * np.sqrt(np.maximum(rel2, 0.0))
    return f


class StubPending:
    def __init__(self, r):
        self.r = r

    def wait(self):
        return self.r


class StubContext:
    """What psfrec.py asks of a Context, answered from the golden file."""

    def __init__(self, tag):
        self.tag = tag
        self.calls = []
        self.offset = 0

    def reconstruct(self, lbda, see, gl, l0, three, h, npsflin=1, masks=None, want_psf=True):
        n, nl = len(see), len(lbda)
        rows = fit_rows_from_columns(self.tag + '_rows', (self.offset + n) * nl)[self.offset * nl:]
        self.calls.append(('reconstruct', n))
        r = {'fit': rows.reshape(n, nl, NFIT), 'psf_sum': np.asarray(G7[self.tag + '_psf_mean']) * n,
             'psf': np.repeat(np.asarray(G7[self.tag + '_psf_mean'])[None], n, axis=0) if want_psf else None}
        return r

    def reconstruct_async(self, lbda, see, gl, l0, three, h, npsflin=1, masks=None, want_psf=True):
        r = self.reconstruct(lbda, see, gl, l0, three, h, npsflin, masks, want_psf)
        self.calls[-1] = ('reconstruct_async', len(see))
        self.offset += len(see)
        return StubPending(r)

    def fit_stamps(self, stamps):
        self.calls.append(('fit_stamps', len(stamps)))
        return fit_rows_from_columns(self.tag + '_mean', len(stamps))

    def abandon(self):
        self.calls.append(('abandon', 0))


def describe_table_hdu(h):
    d = h.data
    return {'type': type(h).__module__ + '.' + type(h).__name__, 'name': h.name,
            'colnames': list(d.columns.names), 'dtypes': {c: str(d[c].dtype.newbyteorder('=')) for c in d.columns.names},
            'shapes': {c: list(d[c].shape) for c in d.columns.names},
            'header_has': {k: (k in h.header) for k in ('SEEING', 'GL', 'L0')},
            'header': {k: float(h.header[k]) for k in ('SEEING', 'GL', 'L0') if k in h.header},
            'values': {c: np.asarray(d[c], dtype=float).tolist() for c in d.columns.names}}


def sparta_input():
    cols = {str(c): np.asarray(G7['table'])[i] for i, c in enumerate(G7['colnames'])}
    tbl = fits.table_to_hdu(ATable(cols))
    tbl.name = 'SPARTA_ATM_DATA'
    return fits.HDUList([fits.PrimaryHDU(), tbl])


out = {'astropy': list(map(lambda m: m is not None, P._astropy()))}

# ---- compute_psf: (astropy Table, cube)
stub = StubContext('mean')
P.get_context = lambda *a, **k: stub
tab, cube = P.compute_psf(np.linspace(490, 930, 4), 1.0, 0.7, 25.0, verbose=False)
out['compute_psf'] = {'type': type(tab).__module__ + '.' + type(tab).__name__, 'colnames': list(tab.colnames),
                      'meta': {k: float(v) for k, v in tab.meta.items()},
                      'shapes': {c: list(np.asarray(tab[c]).shape) for c in tab.colnames},
                      'dtypes': {c: str(np.asarray(tab[c]).dtype) for c in tab.colnames},
                      'cube_shape': list(cube.shape), 'len': len(tab)}

# ---- compute_psf_from_sparta, both modes, small table (one blocking call) and the asynchronous-parts path
for tag, mean in (('mean', True), ('lgs', False)):
    for path in ('one_call', 'async_parts'):
        stub = StubContext(tag)
        P.get_context = lambda *a, **k: stub
        P.PIPELINE_MIN_TASKS = 125 if path == 'one_call' else 2       # (4 / 15 tasks: >= 2 x 2 takes the parts)
        res = P.compute_psf_from_sparta(sparta_input(), lmin=490, lmax=930, nl=4, n_jobs=1, mean_of_lgs=mean,
                                        verbose=False, device=0)
        out['%s_%s' % (tag, path)] = {
            'type': type(res).__module__ + '.' + type(res).__name__,
            'names': [h.name for h in res],
            'hdu_types': [type(h).__name__ for h in res],
            'FIT_ROWS': describe_table_hdu(res['FIT_ROWS']),
            'FIT_MEAN': describe_table_hdu(res['FIT_MEAN']),
            'PSF_MEAN': {'shape': list(res['PSF_MEAN'].data.shape),
                         'max_abs_diff': float(np.abs(res['PSF_MEAN'].data - G7[tag + '_psf_mean']).max())},
            'sparta_copy_rows': int(len(res['SPARTA_ATM_DATA'].data)),
            'calls': stub.calls}
        # the HDUList survives a round trip through a FITS file written by astropy
        p = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'astropy_env_%s_%s.fits' % (tag, path))
        res.writeto(p, overwrite=True)
        with fits.open(p) as back:
            out['%s_%s' % (tag, path)]['roundtrip_names'] = [h.name for h in back]
            out['%s_%s' % (tag, path)]['roundtrip_rows'] = int(len(back['FIT_ROWS'].data))
        os.remove(p)

# ---- create_sparta_table with astropy: a BinTableHDU the reference's own reader accepts
hdu = P.create_sparta_table(nlines=3, seeing=0.9, L0=20, GL=0.6, bad_l0=True)
out['create_sparta_table'] = {'type': type(hdu).__name__, 'name': hdu.name, 'ncols': len(hdu.data.columns.names),
                              'lgs4_l0': np.asarray(hdu.data['LGS4_L0']).tolist()}
json.dump(out, sys.stdout)
