"""A synchronous call (host outputs) split into passes over the two lanes: ms per call against chunk_tasks."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
lb = np.linspace(465, 930, 35)
for dim, n in ((512, 100), (512, 250)):
    see, gl, l0 = synthetic_rows(n)
    three = np.zeros(n, np.uint8)
    ctx = Context(dim=dim, pixscale=grid_pixscale(dim))
    ref = None
    for ch in (n, 0, (n + 1) // 2, n, 0, (n + 1) // 2, n, 0):
        ctx.set_option('chunk_tasks', ch)
        for _ in range(5):
            r = ctx.reconstruct(lb, see, gl, l0, three, (100, 10000), want_psf=False)
        K = 100
        t0 = time.perf_counter()
        for _ in range(K):
            r = ctx.reconstruct(lb, see, gl, l0, three, (100, 10000), want_psf=False)
        dt = (time.perf_counter() - t0) / K
        if ref is None:
            ref = r
        same = bool(np.array_equal(ref['fit'], r['fit']))
        dsum = float(np.max(np.abs(ref['psf_sum'] - r['psf_sum'])) / np.max(np.abs(ref['psf_sum'])))
        print('dim %d rows %d chunk_tasks %d: %.3f ms per call, %.2f M PSFs/s, fit identical %s, psf_sum rel diff %.1e'
              % (dim, n, ch, dt * 1e3, n * 35 / dt / 1e6, same, dsum), flush=True)
    ctx.close()
