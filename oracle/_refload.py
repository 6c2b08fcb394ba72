"""Loader for the upstream reference (TEST INFRASTRUCTURE, runs only in the build container).

Imports /root/reference/muse_psfr/psfrec.py *unmodified* under /opt/conda/bin/python3.9
(the only interpreter here with astropy).  `mpdaf` is not installed anywhere, so an inert
stand-in module is seeded: psfrec.py only needs `mpdaf.obj.Cube` to exist at import time and
nothing in the numerical path before the Moffat fit calls it.  The Moffat fit itself is NOT
reference code (it lives in mpdaf); goldens for it come from the scipy
restatement in oracle/psfr_oracle.py -- the stand-in Cube yields planes whose `moffat_fit` is
that restatement, so that the reference's own compute_psf / compute_psf_from_sparta run end to
end (section `lgs` of make_golden.py).

`load_reference(dim=None, pixscale=None)` returns a module object.  With dim/pixscale given the
reference *source text* is patched in memory (the hard-coded locals psfrec.py:659, :899, :955)
and exec'd -- still the reference's arithmetic, never written to disk.
"""
import importlib.util
import sys
import types

import numpy as np

REF = '/root/reference/muse_psfr/psfrec.py'


class _FitResult:
    """What psfrec.py:866-870 reads from mpdaf's fit object (its __dict__ becomes a table row)."""


class _Image:
    """One plane of the stand-in Cube.  moffat_fit is the scipy/MINPACK restatement of
    oracle/psfr_oracle.py: the Moffat fit is NOT reference code (mpdaf)."""

    def __init__(self, data):
        self.data = data

    def moffat_fit(self, unit_center=None, unit_fwhm=None, circular=True, fit_back=False,
                   verbose=False):
        import psfr_oracle as O
        assert circular and not fit_back and unit_center is None and unit_fwhm is None
        f = O.moffat_fit(self.data, 0.2, errors=True)
        r = _FitResult()
        fw = f['fwhm'] / 0.2                        # pixels: psfrec.py:868 multiplies by 0.2
        r.center = f['center'].copy()
        r.flux = f['flux']
        r.fwhm = np.array([fw, fw])
        r.cont, r.n, r.rot, r.peak = 0.0, f['n'], 0.0, f['peak']
        # mpdaf's error recipe as restated in psfr_oracle.moffat_fit (cov_x * chi2 / dof)
        # (attribute order = mpdaf's Moffat2D: it becomes the column order of the reference's table)
        r.err_center = f['err_center'].copy()
        r.err_flux = f['err_flux']
        r.err_fwhm = np.array([f['err_fwhm'], f['err_fwhm']]) / 0.2      # psfrec.py:869 multiplies by 0.2
        r.err_cont = 0.0
        r.err_n = f['err_n']
        r.err_rot = 0.0
        r.err_peak = f['err_peak']
        r.ima = None
        return r


def _seed():
    # astropy 4.3.1 references names removed from numpy >= 1.23
    if not hasattr(np, 'asscalar'):
        np.asscalar = lambda a: a.item()
    if not hasattr(np, 'alen'):
        np.alen = len
    if 'mpdaf' not in sys.modules:
        mp = types.ModuleType('mpdaf')
        obj = types.ModuleType('mpdaf.obj')

        class Cube:  # stand-in for `from mpdaf.obj import Cube`: planes with the scipy Moffat fit
            def __init__(self, data=None, copy=False):
                self.data = data

            def __iter__(self):
                return (_Image(p) for p in self.data)
        obj.Cube = Cube
        mp.obj = obj
        sys.modules['mpdaf'] = mp
        sys.modules['mpdaf.obj'] = obj


def load_reference(dim=None, pixscale=None):
    _seed()
    if dim is None and pixscale is None:
        spec = importlib.util.spec_from_file_location('ref_psfrec', REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    src = open(REF).read()
    n0 = src.count('pixscale = 0.2')
    assert n0 == 2, n0
    if pixscale is not None:
        src = src.replace('pixscale = 0.2', 'pixscale = %r' % float(pixscale))
    if dim is not None:
        assert src.count('dim=1280, three_lgs_mode=three_lgs_mode') == 1
        src = src.replace('dim=1280, three_lgs_mode=three_lgs_mode',
                          'dim=%d, three_lgs_mode=three_lgs_mode' % int(dim))
    mod = types.ModuleType('ref_psfrec_patched')
    mod.__file__ = REF  # coeffL0.fits is located relative to __file__
    exec(compile(src, REF, 'exec'), mod.__dict__)
    return mod
