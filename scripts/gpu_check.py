"""Ad-hoc GPU-vs-oracle comparison, stage by stage (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import psfr_oracle as O
from muse_psfr_amd import Context, grid_pixscale

def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())

dims = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [128, 256]
precs = sys.argv[2].split(',') if len(sys.argv) > 2 else ['f64', 'mixed']
H = (100, 10000)
for dim in dims:
    ps = grid_pixscale(dim)
    lb = np.array([465., 600., 930.]) if dim != 1280 else np.array([490., 700., 930.])
    cases = [(1.0, 0.7, 25.0, 0), (1.5, 0.3, 10.0, 1)]
    for npl in (1, 3) if dim <= 256 else (1,):
        # oracle with the exact masks (what the library does when masks=None)
        tabs = {g: O.ao_tables(H, bool(g), npl, exact_masks=True) for g in (0, 1)}
        for prec in precs:
            t = time.time()
            ctx = Context(dim=dim, pixscale=ps, precision=prec)
            tc = time.time() - t
            see = np.array([c[0] for c in cases]); gl = np.array([c[1] for c in cases])
            l0 = np.array([c[2] for c in cases]); three = np.array([c[3] for c in cases])
            t = time.time()
            r = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
            tr = time.time() - t
            ndir = npl * npl
            tab = ctx.debug_fetch('ao_tables', (2, ndir, 3, 80, 80))
            tel = ctx.debug_fetch('tel', (dim // 2 + 1, dim))
            d0 = ctx.debug_fetch('dphi0', (len(cases), ndir, dim // 2 + 1, dim))
            pre = ctx.debug_fetch('pre', (len(cases), lb.size, 40, 40))
            print('N=%d npsflin=%d %s: create %.2fs reconstruct %.3fs' % (dim, npl, prec, tc, tr))
            for g in (0, 1):
                T, noise = tabs[g]
                T = T.copy(); T[..., 0, 0] = 0      # psfrec.py:490 (the library zeroes the table)
                ot = np.stack([np.swapaxes(T[0], -1, -2), np.swapaxes(T[1], -1, -2),
                               np.swapaxes(noise, -1, -2)], axis=1)
                print('   ao_tables geom%d' % g, rel(tab[g], ot))
            otel = O.telescope_otf(dim) * dim * dim
            print('   tel', rel(tel, otel[:, :dim // 2 + 1].T))
            for k, (s, g_, l, th) in enumerate(cases):
                psd = O.residual_psd([g_, 1 - g_], H, s, l, npl, dim, bool(th), tables=tabs[th])
                od0 = np.array([O.structure_function0(p) for p in psd])
                print('   case%d dphi0' % k, rel(d0[k], np.swapaxes(od0, -1, -2)[:, :dim // 2 + 1, :]))
                opre = O.psf_stamps_refshaped(psd, lb, 40, ps)
                print('   case%d pre  ' % k, rel(pre[k], opre))
                ofin = O.convolve_final_psf(lb, s, g_, l, opre, ps)
                print('   case%d fin  ' % k, rel(r['psf'][k], ofin))
                ofit = O.fit_psf_cube(ofin, ps)
                gf = r['fit'][k]
                print('   case%d fit  dfwhm %.2e dbeta %.2e dpeak %.2e dcen %.2e  it %s st %s' % (
                    k, np.abs(gf[:, 5] * ps - ofit[:, 3]).max(), np.abs(gf[:, 4] - ofit[:, 4]).max(),
                    np.abs(gf[:, 0] / ofit[:, 0] - 1).max(), np.abs(gf[:, 1:3] - ofit[:, 1:3]).max(),
                    gf[:, 7].tolist(), gf[:, 14].tolist()))
            print('   sum vs psf', rel(r['psf_sum'], r['psf'].sum(axis=0)))
            ctx.close()
