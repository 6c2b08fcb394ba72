"""Loader for the upstream reference (TEST INFRASTRUCTURE, runs only in the build container).

Imports /root/reference/muse_psfr/psfrec.py *unmodified* under /opt/conda/bin/python3.9
(the only interpreter here with astropy).  `mpdaf` is not installed anywhere, so an inert
stand-in module is seeded: psfrec.py only needs `mpdaf.obj.Cube` to exist at import time and
nothing in the numerical path before the Moffat fit calls it.  The Moffat fit itself is NOT
reference code (it lives in mpdaf); goldens for it come from the scipy
restatement in oracle/psfr_oracle.py.

`load_reference(dim=None, pixscale=None)` returns a module object.  With dim/pixscale given the
reference *source text* is patched in memory (the hard-coded locals psfrec.py:659, :899, :955)
and exec'd -- still the reference's arithmetic, never written to disk.
"""
import importlib.util
import sys
import types

import numpy as np

REF = '/root/reference/muse_psfr/psfrec.py'


def _seed():
    # astropy 4.3.1 references names removed from numpy >= 1.23
    if not hasattr(np, 'asscalar'):
        np.asscalar = lambda a: a.item()
    if not hasattr(np, 'alen'):
        np.alen = len
    if 'mpdaf' not in sys.modules:
        mp = types.ModuleType('mpdaf')
        obj = types.ModuleType('mpdaf.obj')

        class Cube:  # inert: only has to exist for `from mpdaf.obj import Cube`
            def __init__(self, data=None, copy=False):
                self.data = data
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
