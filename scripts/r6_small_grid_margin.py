"""Round 6, VERDICT r5 #7: the row behind `dim 256 nl 5 rows 257 ... beta 8.0e-05` of scripts/robust_sweep.py.
Mixed vs f64 vs the oracle on that row, and the sensitivity of n to a perturbation of the stamp
    sens_n = n^2 sqrt(cov[eta, eta]) * peak = err_n * peak / sqrt(chi2 / dof)
(the change of n per unit of iid pixel noise relative to the peak, from the covariance the fit kernel already forms)
as the principled measure of "well-posed".  Writes gpurun_out/r6_small_grid_margin.json."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from muse_psfr_amd import Context, grid_pixscale, synthetic_rows
import psfr_oracle as O
H = (100, 10000)
out = {}
for dim in (256, 128, 512):
    ps = grid_pixscale(dim)
    nl, rows = 5, 257
    lb = np.linspace(465.0, 930.0, nl)
    lb = lb[np.random.default_rng(nl).permutation(nl)]
    see, gl, l0 = synthetic_rows(rows)
    three = (np.arange(rows) % 3 == 1).astype(np.uint8)
    res = {}
    for p in ('mixed', 'f64'):
        c = Context(dim=dim, pixscale=ps, precision=p)
        res[p] = c.reconstruct(lb, see, gl, l0, three, H)
        c.close()
    a, b = res['mixed']['fit'], res['f64']['fit']
    dof = 1600 - 5
    sens = b[..., 12] * b[..., 0] / np.sqrt(np.maximum(b[..., 6], 1e-300) / dof)
    dn = np.abs(a[..., 4] - b[..., 4])
    dstamp = np.abs(res['mixed']['psf'] - res['f64']['psf']).max(axis=(2, 3)) / res['f64']['psf'].max(axis=(2, 3))
    old_well = (b[..., 14] == 0) & (b[..., 4] < 20) & (b[..., 5] > 2.5)
    k = np.unravel_index(np.argmax(np.where(old_well, dn, 0)), dn.shape)
    print('dim %d: worst well-posed (old rule) stamp: row %d lambda %.1f nm: seeing %.3f GL %.3f L0 %.2f three %d' % (
        dim, k[0], lb[k[1]], see[k[0]], gl[k[0]], l0[k[0]], three[k[0]]))
    print('   f64 fit: n %.5f fwhm %.4f px peak %.3e chi2 %.3e err_n %.3e  |dn| mixed-f64 %.2e  stamp diff %.2e of peak  sens_n %.1f' % (
        b[k][4], b[k][5], b[k][0], b[k][6], b[k][12], dn[k], dstamp[k], sens[k]))
    # dn against the prediction sens_n * (stamp difference)
    ratio = dn / np.maximum(sens * dstamp, 1e-300)
    print('   over all %d stamps with n < 20: max |dn| / (sens_n x stamp diff) = %.2f (a bound would be ~ sqrt(1600) = 40 in the worst case, ~1 for noise-like differences)' % (
        (b[..., 4] < 20).sum(), ratio[b[..., 4] < 20].max()))
    for thr in (30, 100, 300, 1000):
        w = (b[..., 14] == 0) & (sens < thr)
        print('   sens_n < %5d: %4d of %d stamps, worst |dn| %.2e, worst |dfwhm| %.2e arcsec' % (
            thr, w.sum(), w.size, dn[w].max(initial=0), (np.abs(a[..., 5] - b[..., 5])[w].max(initial=0)) * ps))
    # the oracle on the worst row
    tabs = O.ao_tables(H, bool(three[k[0]]), 1, exact_masks=True)
    ofit, ofin = O.compute_psf(lb, see[k[0]], gl[k[0]], l0[k[0]], 1, H, bool(three[k[0]]), dim=dim, pixscale=ps, tables=tabs)
    on = ofit[k[1], 4]
    sm = np.abs(res['mixed']['psf'][k[0]] - ofin).max() / ofin.max()
    sf = np.abs(res['f64']['psf'][k[0]] - ofin).max() / ofin.max()
    # MINPACK on the GPU's own stamps: how much of the difference is the stamp, how much the fit
    mfit_m = O.moffat_fit(res['mixed']['psf'][k], ps, errors=True)
    mfit_f = O.moffat_fit(res['f64']['psf'][k], ps, errors=True)
    print('   oracle n %.6f | GPU mixed %.6f (d %.1e) | GPU f64 %.6f (d %.1e) | MINPACK on the mixed stamp %.6f, on the f64 stamp %.6f' % (
        on, a[k][4], abs(a[k][4] - on), b[k][4], abs(b[k][4] - on), mfit_m['n'], mfit_f['n']))
    print('   stamps of the row vs the oracle: mixed %.1e, f64 %.1e of the peak' % (sm, sf))
    out['dim%d' % dim] = {'row': int(k[0]), 'lambda_nm': float(lb[k[1]]), 'seeing': float(see[k[0]]), 'GL': float(gl[k[0]]),
                          'L0': float(l0[k[0]]), 'n_f64': float(b[k][4]), 'fwhm_px': float(b[k][5]), 'sens_n': float(sens[k]),
                          'dn_mixed_vs_f64': float(dn[k]), 'dn_mixed_vs_oracle': float(abs(a[k][4] - on)),
                          'dn_f64_vs_oracle': float(abs(b[k][4] - on)), 'stamp_diff_mixed_vs_f64': float(dstamp[k]),
                          'n_minpack_on_mixed_stamp': float(mfit_m['n']), 'n_minpack_on_f64_stamp': float(mfit_f['n']),
                          'n_oracle': float(on),
                          'worst_dn_by_sens_threshold': {str(t): float(dn[(b[..., 14] == 0) & (sens < t)].max(initial=0)) for t in (30, 100, 300, 1000)},
                          'count_by_sens_threshold': {str(t): int(((b[..., 14] == 0) & (sens < t)).sum()) for t in (30, 100, 300, 1000)},
                          'stamps': int(dn.size)}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r6_small_grid_margin.json'), 'w'), indent=1)
