import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
see, gl, l0 = synthetic_rows(100)
lb = np.linspace(465, 930, 35)
ctx = Context(dim=512, pixscale=grid_pixscale(512))
r = ctx.reconstruct(lb, see, gl, l0, np.zeros(100, np.uint8), (100, 10000))
it = r['fit'][:, :, 7]
print('iterations: min %d median %d mean %.1f max %d' % (it.min(), np.median(it), it.mean(), it.max()))
print(np.bincount(it.astype(int).ravel()))
print('beta range', r['fit'][:, :, 4].min(), r['fit'][:, :, 4].max(), 'fwhm px', r['fit'][:, :, 5].min(), r['fit'][:, :, 5].max())
b = r['fit'][:, :, 4]
print('beta percentiles 5/25/50/75/95:', np.percentile(b, [5, 25, 50, 75, 95]))
