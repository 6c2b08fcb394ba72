"""K_DPHI_SERIES_Q against K_DPHI_SERIES: option stage_a_queue 0 / 1 / 2 on several grids, both precisions.
Mode 1 must be bit-identical in everything; mode 2 (lines skipped) in the fits to 1e-7 and prints what it skipped."""
import sys, numpy as np
sys.path.insert(0, '.')
import muse_psfr_amd as M
H = (100, 10000)
modes = [int(x) for x in sys.argv[1:]] or [1]
for dim, npl, n in ((512, 1, 37), (256, 3, 9), (1024, 1, 11), (1280, 1, 13), (128, 1, 5), (512, 2, 6)):
    for prec in ('mixed', 'f64'):
        see, gl, l0 = M.synthetic_rows(n, seed=31)
        three = (np.arange(n) % 4 == 1).astype(np.uint8)
        lb = np.linspace(490, 930, 6)
        ps = M.grid_pixscale(dim) if dim != 1280 else 0.2
        out = {}
        for mode in [0] + modes:
            c = M.Context(dim=dim, pixscale=ps, precision=prec)
            c.set_option('stage_a', 2)
            c.set_option('stage_a_queue', mode)
            for rep in range(2):
                r = c.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
            d = c.debug_fetch('dphi0', (n, npl * npl, dim // 2 + 1, dim))
            out[mode] = (r, d)
            c.close()
        for mode in modes:
            (ra, da), (rb, db) = out[0], out[mode]
            if mode == 1:
                ok = all(np.array_equal(ra[k], rb[k]) for k in ('psf', 'fit', 'psf_sum')) and np.array_equal(da, db)
                print('dim %4d npsflin %d %-5s queue: bit-identical %s' % (dim, npl, prec, ok), flush=True)
            else:
                skipped = (db >= 1e29).all(axis=-1)              # whole lines
                same = np.array_equal(da[~skipped], db[~skipped])
                well = ra['fit'][..., 14] == 0
                dfit = np.abs(ra['fit'][..., 4:6] - rb['fit'][..., 4:6])[well].max() if well.any() else 0
                dst = np.abs(ra['psf'] - rb['psf']).max() / ra['psf'].max()
                print('dim %4d npsflin %d %-5s skip: lines skipped %.3f, kept lines identical %s, stamps %.1e, fit (n, fwhm px) %.1e' % (
                    dim, npl, prec, skipped.mean(), same, dst, dfit), flush=True)
