"""Odd shapes through the C ABI: every combination must run, give finite results, and the mixed mode must
agree with the f64 mode on the fits of well-posed stamps (1e-4 on fwhm / beta) and on the stamps (1e-5 of peak)."""
import itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, grid_pixscale, synthetic_rows
bad = 0
for dim in (128, 256, 512, 1024, 1280):
    ctxs = {p: Context(dim=dim, pixscale=grid_pixscale(dim), precision=p) for p in ('mixed', 'f64')}
    for nl, rows, npl in ((1, 1, 1), (2, 3, 1), (7, 5, 2), (36, 2, 1), (71, 3, 1), (128, 1, 1), (5, 257, 1), (3, 4, 3)):
        if dim >= 1024 and (rows > 100 or nl > 71 or npl > 2):
            continue
        lo = 490.0 if dim == 1280 else 465.0
        lb = np.linspace(lo, 930.0, nl) if nl > 1 else np.array([700.0])
        lb = lb[np.random.default_rng(nl).permutation(nl)]             # any order
        see, gl, l0 = synthetic_rows(rows)
        three = (np.arange(rows) % 3 == 1).astype(np.uint8)
        res = {}
        try:
            for p, c in ctxs.items():
                t = time.perf_counter()
                res[p] = c.reconstruct(lb, see, gl, l0, three, (100, 10000), npsflin=npl)
                dt = time.perf_counter() - t
        except Exception as e:
            print('FAIL dim %d nl %d rows %d npsflin %d: %s' % (dim, nl, rows, npl, e)); bad += 1; continue
        a, b = res['mixed'], res['f64']
        ok = all(np.isfinite(r[k]).all() for r in (a, b) for k in ('psf', 'fit', 'psf_sum'))
        ds = np.abs(a['psf'] - b['psf']).max() / b['psf'].max()
        # well-posed: status 0 in both precisions (round 6: the fit kernel flags stamps that do not pin (fwhm, n) to the
        # tolerance -- MPSFR_FIT_ILL_CONDITIONED, from its own covariance -- instead of the ad-hoc  beta < 20 & fwhm > 2.5 px)
        well = (b['fit'][..., 14] == 0) & (a['fit'][..., 14] == 0)
        df = np.abs(a['fit'][..., 5] - b['fit'][..., 5])[well].max() * 0.2 if well.any() else 0.0
        dn = np.abs(a['fit'][..., 4] - b['fit'][..., 4])[well].max() if well.any() else 0.0
        flag = '' if ok and ds < 1e-5 and df < 1e-4 and dn < 1e-4 else '   <-- CHECK'
        bad += bool(flag)
        print('dim %4d nl %3d rows %3d npsflin %d: finite %s  stamps %.1e  fwhm %.1e  beta %.1e  (%d well-posed of %d)%s' % (
            dim, nl, rows, npl, ok, ds, df, dn, well.sum(), well.size, flag), flush=True)
    for c in ctxs.values():
        c.close()
print('issues:', bad)
