"""Mixed-precision fits against the f64 mode of the same library on the bench workload (quick
proxy for the oracle comparison: the f64 mode agrees with the oracle to 1e-8)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
see, gl, l0 = synthetic_rows(n)
lb = np.linspace(465, 930, 35)
ps = grid_pixscale(512)
res = {}
for prec in ('mixed', 'f64'):
    ctx = Context(dim=512, pixscale=ps, precision=prec)
    res[prec] = ctx.reconstruct(lb, see, gl, l0, np.zeros(n, np.uint8), (100, 10000))['fit']
    ctx.close()
a, b = res['mixed'], res['f64']
print('max |dfwhm| arcsec %.3e   max |dbeta| %.3e   mean iterations %.2f (f64 %.2f)' % (
    np.abs(a[:, :, 5] - b[:, :, 5]).max() * ps, np.abs(a[:, :, 4] - b[:, :, 4]).max(),
    a[:, :, 7].mean(), b[:, :, 7].mean()))
print('iteration histogram (mixed):', np.bincount(a[:, :, 7].astype(int).ravel()))
