"""Series + patch form of stage A against the full-size transforms: D_phi0 difference and per-kernel
times.  usage: python scripts/r4_stage_a.py [dims] [precs] [rows]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muse_psfr_amd import Context, grid_pixscale

dims = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [512]
precs = sys.argv[2].split(',') if len(sys.argv) > 2 else ['mixed']
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100
prof_only = len(sys.argv) > 4     # a 5th argument: only the series form, only full-size calls (for rocprofv3)
H = (100, 10000)
rng = np.random.default_rng(355)
see = rng.uniform(0.4, 1.6, rows); gl = rng.uniform(0.3, 0.95, rows); l0 = rng.uniform(9, 29, rows)
l0[0] = 7.0; l0[1] = 1000.0
three = np.zeros(rows, np.uint8)
for dim in dims:
    ps = 0.2 if dim == 1280 else grid_pixscale(dim)
    lb = np.linspace(490, 930, 35) if dim == 1280 else np.linspace(465, 930, 35)
    for prec in precs:
        for npl in (1,):
            res = {}
            for mode in ((1,) if prof_only else (0, 1)):
                ctx = Context(dim=dim, pixscale=ps, precision=prec)
                ctx.set_option('stage_a', mode)
                nf = min(rows, 4)
                d0 = None
                if not prof_only:
                    ctx.reconstruct(lb, see[:nf], gl[:nf], l0[:nf], three[:nf], H, npsflin=npl)
                    d0 = ctx.debug_fetch('dphi0', (nf, npl * npl, dim // 2 + 1, dim))
                r = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
                ctx.set_option('profile', 1)
                ctx.profile_reset()
                for _ in range(5):
                    ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
                prof = ctx.profile()
                ctx.close()
                res[mode] = (d0, r, prof)
                print('N=%d %s stage_a=%d: ' % (dim, prec, mode) + ' '.join('%s %.1f' % (k, 1e3 * v[0] / max(v[1], 1)) for k, v in prof.items() if v[1]), flush=True)
            if prof_only:
                continue
            d_leg, d_ser = res[0][0], res[1][0]
            for k in range(d_leg.shape[0]):
                print('   row %d (L0 %.1f): |dD| / max D = %.3e   max D %.4g' % (k, l0[k], np.abs(d_ser[k] - d_leg[k]).max() / np.abs(d_leg[k]).max(), np.abs(d_leg[k]).max()))
            f0, f1 = res[0][1]['fit'], res[1][1]['fit']
            print('   fits: |d fwhm px| %.3e  |d beta| %.3e   stamps %.3e' % (np.abs(f0[..., 5] - f1[..., 5]).max(), np.abs(f0[..., 4] - f1[..., 4]).max(), np.abs(res[0][1]['psf'] - res[1][1]['psf']).max() / res[0][1]['psf'].max()))
