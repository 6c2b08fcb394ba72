"""Golden-vector generator.  TEST INFRASTRUCTURE; runs ONLY in the build container:

    /opt/conda/bin/python3.9 -B oracle/make_golden.py [section ...]

It executes the real reference (/root/reference, via oracle/_refload.py), asserts that the
NumPy restatement oracle/psfr_oracle.py agrees with it stage by stage, and writes small .npz
fixtures (inputs + the reference's outputs) to tests/golden/.  Nothing here travels as code the
GPU box needs; the fixtures are data.

Sections: masks native grids sparta lgs
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
from _refload import load_reference  # noqa: E402
import psfr_oracle as O  # noqa: E402
from muse_psfr_amd.synthetic import synthetic_rows, grid_pixscale  # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
H = (100, 10000)
CASES = [(1.0, 0.7, 25.0), (1.5, 0.3, 10.0), (0.5, 0.9, 29.0), (2.0, 0.5, 20.0)]
LB5 = np.array([490., 500., 700., 900., 930.])


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def check(name, a, b, tol=1e-12):
    r = rel(a, b)
    print('  %-34s rel.err oracle vs reference = %.2e' % (name, r), flush=True)
    assert r < tol, (name, r)


def ref_masks():
    """The cut-off masks as THIS interpreter's NumPy evaluates psfrec.py:257/:435."""
    f, f_x, f_y = O._ao_freqs()
    fc = 1.5
    ge = (f != 0) & (np.abs(f_x) >= fc) | (np.abs(f_y) >= fc)
    gt = (f != 0) & (np.abs(f_x) > fc) | (np.abs(f_y) > fc)
    return ge, gt


def section_masks(ref):
    """G1: cut-off masks + the reference's dsp4muse output for two inputs x two geometries."""
    ge, gt = ref_masks()
    out = dict(mask_rec=np.packbits(ge), mask_res=np.packbits(gt),
               numpy_version=np.array(np.__version__))
    for tag, three, npl in (('4lgs', False, 1), ('3lgs', True, 1), ('4lgs_n3', False, 3)):
        for ci, (see, gl, l0) in enumerate(CASES[:2]):
            cn2 = np.array([gl, 1 - gl])
            r0 = ref.seeing2r01(see, 0.5, 0.)
            poslgs = O.lgs_positions(three)
            dirperf = ref.direction_perf(npl)
            h = np.array(H)
            vent = np.full_like(h, 12.5)
            dsp = ref.dsp4muse(8., 40, 80, cn2, h, l0, r0, 1, 1., vent,
                               np.array([0.628163, -0.326497]), 'LSE', 24., 24., 1000., 2.5,
                               1.0, 0.5, poslgs, dirperf)
            mine = O.ao_zone_psd(cn2, H, l0, r0, three, npl)
            check('dsp4muse %s case%d' % (tag, ci), mine, dsp, 1e-13)
            if npl == 1:
                out['dsp_%s_c%d' % (tag, ci)] = dsp
            else:   # 9 directions: sums + 16 sampled pixels per direction
                idx = np.array([1, 5, 23, 24, 25, 40, 55, 56, 57, 79])
                out['dsp_%s_c%d_sum' % (tag, ci)] = dsp.sum(axis=(1, 2))
                out['dsp_%s_c%d_samp' % (tag, ci)] = dsp[:, idx][:, :, idx]
                out['samp_idx'] = idx
    out['cases'] = np.array(CASES[:2])
    np.savez_compressed(os.path.join(OUT, 'g1_ao_zone.npz'), **out)


def run_ref(ref, lb, see, gl, l0, npl, three, dim):
    psd = ref.simul_psd_wfm([gl, 1 - gl], H, see, l0, zenith=0., npsflin=npl, dim=dim,
                            three_lgs_mode=three, verbose=False)
    p = psd[0] if npl == 1 else psd
    pre = ref.psf_muse(p, lb)
    fin = ref.convolve_final_psf(lb, see, gl, l0, pre)
    return psd, pre, fin


def section_native(ref):
    """G2-G4: native 1280 grid, 4 inputs (+ npsflin=3, + 3-LGS) at 5 wavelengths."""
    out = dict(lbda=LB5, h=np.array(H))
    runs = [(c, 1, False) for c in CASES] + [(CASES[0], 3, False), (CASES[0], 1, True),
                                             (CASES[1], 3, True)]
    meta = []
    for k, ((see, gl, l0), npl, three) in enumerate(runs):
        t = time.time()
        psd, pre, fin = run_ref(ref, LB5, see, gl, l0, npl, three, 1280)
        print('run %d: reference %.1fs' % (k, time.time() - t), flush=True)
        opsd = O.residual_psd([gl, 1 - gl], H, see, l0, npl, 1280, three)
        check('psd', opsd, psd)
        opre = O.psf_stamps_refshaped(opsd, LB5)
        check('stamps pre-conv', opre, pre)
        ores = O.psf_stamps_restructured(opsd, LB5)
        check('stamps restructured', ores, pre)
        ofin = O.convolve_final_psf(LB5, see, gl, l0, opre)
        check('stamps final', ofin, fin)
        fit = O.fit_psf_cube(fin)
        meta.append((see, gl, l0, npl, int(three)))
        c = 640
        out['psd_centre_%d' % k] = psd[:, c - 48:c + 48, c - 48:c + 48]
        out['psd_row0_%d' % k] = psd[:, 0, :]
        out['psd_rowc_%d' % k] = psd[:, c, :]
        out['psd_sum_%d' % k] = psd.sum(axis=(1, 2))
        out['pre_%d' % k] = pre
        out['fin_%d' % k] = fin
        out['fit_%d' % k] = fit
        print('   fit', fit[[1, 2, 3]][:, 3:].round(6).tolist(), flush=True)
    out['meta'] = np.array(meta)
    np.savez_compressed(os.path.join(OUT, 'g2_native1280.npz'), **out)


def section_grids(ref_unused):
    """G6: N != 1280 grids via the reference source with its hard-coded locals patched in memory."""
    out = {}
    for dim, nrow in ((128, 2), (256, 3), (512, 4), (1024, 2)):
        ps = grid_pixscale(dim)
        ref = load_reference(dim=dim, pixscale=ps)
        lb = np.array([465., 600., 700., 930.])
        see, gl, l0 = synthetic_rows(nrow)
        rows = [(1.0, 0.7, 25.0)] + [(see[i], gl[i], l0[i]) for i in range(nrow - 1)]
        for k, (s, g, l) in enumerate(rows):
            npl = 3 if (dim == 256 and k == 1) else 1
            three = (k == 2)
            t = time.time()
            psd, pre, fin = run_ref(ref, lb, s, g, l, npl, three, dim)
            opsd = O.residual_psd([g, 1 - g], H, s, l, npl, dim, three)
            check('N=%d psd' % dim, opsd, psd)
            opre = O.psf_stamps_refshaped(opsd, lb, 40, ps)
            check('N=%d pre' % dim, opre, pre)
            ofin = O.convolve_final_psf(lb, s, g, l, opre, ps)
            check('N=%d fin' % dim, ofin, fin)
            key = 'n%d_r%d' % (dim, k)
            out[key + '_in'] = np.array([s, g, l, npl, int(three), ps])
            out[key + '_pre'] = pre
            out[key + '_fin'] = fin
            out[key + '_fit'] = O.fit_psf_cube(fin, ps)
            print('N=%d row %d %.1fs fit@465 %s' % (dim, k, time.time() - t,
                                                     out[key + '_fit'][0, 3:].tolist()), flush=True)
        out['n%d_lbda' % dim] = lb
    np.savez_compressed(os.path.join(OUT, 'g6_grids.npz'), **out)


def section_sparta(ref):
    """G5: 16 synthetic rows + 2 rows with LGS4_L0=150 (3-LGS mode), 35 wavelengths 490-930 nm,
    native grid: per-task fits, the mean PSF and its fit (compute_psf_from_sparta semantics,
    psfrec.py:1041-1113, mean_of_lgs=True)."""
    lb = np.linspace(490, 930, 35)
    see, gl, l0 = synthetic_rows(16)
    three = np.zeros(18, dtype=bool)
    see = np.concatenate([see, [1.0, 0.8]])
    gl = np.concatenate([gl, [0.7, 0.5]])
    l0 = np.concatenate([l0, [25.0, 20.0]])
    three[16:] = True
    from joblib import Parallel, delayed

    def one(i):
        _, pre, fin = run_ref(ref, lb, see[i], gl[i], l0[i], 1, bool(three[i]), 1280)
        return pre, fin
    t = time.time()
    res = Parallel(n_jobs=6, backend='threading')(delayed(one)(i) for i in range(18))
    print('reference: 18 rows x 35 lambda in %.1fs' % (time.time() - t), flush=True)
    fin = np.array([r[1] for r in res])
    fits = np.array([O.fit_psf_cube(f) for f in fin])
    mean = np.mean(fin, axis=0)
    # spot-check the oracle on two rows
    for i in (3, 17):
        _, ofin = O.compute_psf(lb, see[i], gl[i], l0[i], 1, H, bool(three[i]), fit=False)
        check('sparta row %d final' % i, ofin, fin[i])
    np.savez_compressed(os.path.join(OUT, 'g5_sparta18.npz'), lbda=lb, seeing=see, gl=gl, l0=l0,
                        three=three, fit_rows=fits, psf_mean=mean, fit_mean=O.fit_psf_cube(mean),
                        fin_row0=fin[0], fin_row17=fin[17], pre_row0=res[0][0])


def section_lgs(ref):
    """G7: the reference's own compute_psf_from_sparta (psfrec.py:981-1120) end to end on a
    5-row table whose four LGS columns are jittered by +-5 % (BASELINE.json configs[3]), one
    laser with L0 = 100 m (rejected -> 3-LGS mode, psfrec.py:1049-1054) and one row with no valid
    laser at all (skipped, :1056-1060): mean_of_lgs=True (one task per row) and mean_of_lgs=False
    (one task per valid laser, :1071-1076)."""
    from astropy.io import fits
    from astropy.table import Table
    rng = np.random.default_rng(3553)
    nrow = 5
    see, gl, l0 = synthetic_rows(nrow)
    cols = {}
    for k in range(1, 5):
        j = 1 + 0.05 * rng.normal(size=(3, nrow))
        cols['LGS%d_SEEING' % k] = see * j[0]
        cols['LGS%d_TUR_GND' % k] = np.clip(gl * j[1], 0.05, 0.98)
        cols['LGS%d_L0' % k] = np.clip(l0 * j[2], 8.5, 29.5)
    cols['LGS3_L0'][1] = 100.0                      # one rejected laser in row 2
    for k in range(1, 5):
        cols['LGS%d_L0' % k][3] = 150.0             # row 4: nothing valid
    tbl = fits.table_to_hdu(Table(cols))
    tbl.name = 'SPARTA_ATM_DATA'
    out = {'colnames': np.array(list(cols)), 'table': np.array([cols[c] for c in cols]),
           'lmin': 490.0, 'lmax': 930.0, 'nl': 4}
    for tag, mean in (('mean', True), ('lgs', False)):
        t = time.time()
        res = ref.compute_psf_from_sparta(fits.HDUList([fits.PrimaryHDU(), tbl]), lmin=490, lmax=930,
                                          nl=4, n_jobs=1, mean_of_lgs=mean, verbose=False)
        print('reference compute_psf_from_sparta(mean_of_lgs=%s): %.1fs, %d FIT_ROWS rows' % (
            mean, time.time() - t, len(res['FIT_ROWS'].data)), flush=True)
        assert [h.name for h in res] == ['PRIMARY', 'SPARTA_ATM_DATA', 'FIT_ROWS', 'FIT_MEAN', 'PSF_MEAN']
        fr, fm = res['FIT_ROWS'].data, res['FIT_MEAN'].data
        assert fr.columns.names[:11] == ['lbda', 'center', 'flux', 'fwhm', 'n', 'peak', 'err_center', 'err_flux',
                                         'err_fwhm', 'err_n', 'err_peak'], fr.columns.names
        for c in ('lbda', 'fwhm', 'n', 'peak', 'center', 'flux', 'err_center', 'err_flux', 'err_fwhm', 'err_n',
                  'err_peak', 'SEEING', 'GL', 'L0', 'row_idx', 'lgs_idx'):
            out['%s_rows_%s' % (tag, c)] = np.array(fr[c])
        for c in ('lbda', 'fwhm', 'n', 'peak', 'center', 'flux', 'err_center', 'err_flux', 'err_fwhm', 'err_n',
                  'err_peak'):
            out['%s_mean_%s' % (tag, c)] = np.array(fm[c])
        hdr = res['FIT_MEAN'].header
        out['%s_mean_hdr' % tag] = np.array([hdr['SEEING'], hdr['GL'], hdr['L0']])
        out['%s_psf_mean' % tag] = np.array(res['PSF_MEAN'].data)
    np.savez_compressed(os.path.join(OUT, 'g7_sparta_lgs.npz'), **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    secs = sys.argv[1:] or ['masks', 'native', 'grids', 'sparta', 'lgs']
    ref = load_reference()
    for s in secs:
        print('== section', s, flush=True)
        globals()['section_' + s](ref)
