"""CPU: the oracle (oracle/psfr_oracle.py) against the golden vectors captured from the real
reference (oracle/make_golden.py) and against the reference's own published known answers."""
import numpy as np
import pytest

import psfr_oracle as O
from conftest import H, rel_err


def test_cutoff_masks_are_platform_dependent_only_on_the_boundary(ref_masks):
    """The golden masks differ from the exact rule only on the |k| = 24 lines (fc = 24/16)."""
    k = np.abs(np.fft.fftfreq(80, 1 / 80).astype(int))
    exact_rec = (k[:, None] >= 24) | (k[None, :] >= 24)
    exact_res = (k[:, None] > 24) | (k[None, :] > 24)
    on_line = (k[:, None] == 24) | (k[None, :] == 24)
    assert not np.any((ref_masks[0] != exact_rec) & ~on_line)
    assert not np.any((ref_masks[1] != exact_res) & ~on_line)
    assert 0 < np.sum(ref_masks[0] != exact_rec) < 160


@pytest.mark.parametrize('tag,three,npl', [('4lgs', False, 1), ('3lgs', True, 1)])
def test_ao_zone_psd_matches_reference_dsp4muse(golden, ref_masks, tag, three, npl):
    g = golden('g1_ao_zone')
    tabs = O.ao_tables(H, three, npl, masks=ref_masks)
    for ci, (see, gl, l0) in enumerate(g['cases']):
        d = O.ao_zone_psd(np.array([gl, 1 - gl]), H, l0, O.seeing_to_r0(see), three, npl, tables=tabs)
        assert rel_err(d, g['dsp_%s_c%d' % (tag, ci)]) < 1e-13


def test_ao_zone_psd_nine_directions(golden, ref_masks):
    g = golden('g1_ao_zone')
    idx = g['samp_idx']
    tabs = O.ao_tables(H, False, 3, masks=ref_masks)
    for ci, (see, gl, l0) in enumerate(g['cases']):
        d = O.ao_zone_psd(np.array([gl, 1 - gl]), H, l0, O.seeing_to_r0(see), False, 3, tables=tabs)
        np.testing.assert_allclose(d.sum(axis=(1, 2)), g['dsp_4lgs_n3_c%d_sum' % ci], rtol=1e-13)
        np.testing.assert_allclose(d[:, idx][:, :, idx], g['dsp_4lgs_n3_c%d_samp' % ci], rtol=1e-12,
                                   atol=1e-300)


@pytest.mark.parametrize('run', [0, 5])
def test_native_grid_against_reference(golden, ref_masks, run):
    """N = 1280 (the reference's hard-coded grid): PSD, stamps before/after convolution."""
    g = golden('g2_native1280')
    see, gl, l0, npl, three = g['meta'][run]
    npl, three = int(npl), bool(three)
    lb = g['lbda']
    tabs = O.ao_tables(H, three, npl, masks=ref_masks)
    psd = O.residual_psd([gl, 1 - gl], H, see, l0, npl, 1280, three, tables=tabs)
    c = 640
    assert rel_err(psd[:, c - 48:c + 48, c - 48:c + 48], g['psd_centre_%d' % run]) < 1e-13
    assert rel_err(psd[:, 0, :], g['psd_row0_%d' % run]) < 1e-13
    assert rel_err(psd[:, c, :], g['psd_rowc_%d' % run]) < 1e-13
    np.testing.assert_allclose(psd.sum(axis=(1, 2)), g['psd_sum_%d' % run], rtol=1e-12)
    sel = [1, 3]          # 500 and 900 nm
    pre = O.psf_stamps_refshaped(psd, lb[sel])
    assert rel_err(pre, g['pre_%d' % run][sel]) < 1e-12
    pre2 = O.psf_stamps_restructured(psd, lb[sel])
    assert rel_err(pre2, g['pre_%d' % run][sel]) < 1e-12
    fin = O.convolve_final_psf(lb[sel], see, gl, l0, pre)
    assert rel_err(fin, g['fin_%d' % run][sel]) < 1e-12


@pytest.mark.parametrize('dim,npl', [(128, 1), (128, 3), (256, 1)])
def test_split_fp16_contraction_model(dim, npl):
    """The identities and the arithmetic the matrix-core kernel rests on (otf_mfma.hip), on the CPU:
    stamps from the half plane and the 21 distinct samples per direction (P +- Q), every operand as
    two fp16 halves with fp32 accumulation.  Scaled out of the fp16 subnormal range the model sits
    at fp32 level against the reference-shaped oracle; with the tables unscaled (low halves flushed
    by the matrix cores) it is an order of magnitude worse -- the trap DESIGN.md section 7 records."""
    ps = 0.2 * dim / 1344
    lb = np.array([480.0, 700.0, 925.0])
    psd = O.residual_psd([0.7, 0.3], H, 0.6, 25.0, npl, dim, False)
    ref = O.psf_stamps_refshaped(psd, lb, 40, ps)
    peak = ref.max(axis=(1, 2), keepdims=True)
    good = O.psf_stamps_contraction_fp16(psd, lb, 40, ps)
    bad = O.psf_stamps_contraction_fp16(psd, lb, 40, ps, otf_shift=11, tab_shift=0)
    eg, eb = (np.abs(good - ref) / peak).max(), (np.abs(bad - ref) / peak).max()
    assert eg < 3e-6, eg
    assert eb > 5 * eg, (eg, eb)
    np.testing.assert_allclose(good.sum(axis=(1, 2)), 1.0, rtol=1e-6)


def test_fit_oracle_reproduces_the_reference_known_answers(golden):
    """test_psfrec.py:121-127: LBDA 5000 7000 9000 / FWHM 0.85 0.73 0.62 / BETA 2.73 2.55 2.23 for
    (seeing 1.0, GL 0.7, L0 25), printed with two decimals; centre 20 (test_psfrec.py:28)."""
    g = golden('g2_native1280')
    assert tuple(g['meta'][0][:3]) == (1.0, 0.7, 25.0)
    fit = O.fit_psf_cube(g['fin_0'][[1, 2, 3]])
    assert ['%.2f' % v for v in fit[:, 3]] == ['0.85', '0.73', '0.62']
    assert ['%.2f' % v for v in fit[:, 4]] == ['2.73', '2.55', '2.23']
    np.testing.assert_allclose(fit[:, 1:3], 20, atol=1e-3)
    np.testing.assert_allclose(fit, g['fit_0'][[1, 2, 3]], rtol=1e-7)


def test_three_lgs_known_answer(golden):
    """test_psfrec.py:88-90: fwhm 0.86 at 502.9 nm in 3-laser mode; here at 500 nm."""
    g = golden('g2_native1280')
    assert tuple(g['meta'][5]) == (1.0, 0.7, 25.0, 1, 1)
    assert abs(g['fit_5'][1, 3] - 0.86) < 1e-2


@pytest.mark.parametrize('dim,rows', [(128, 2), (256, 3), (512, 2)])
def test_small_grids_against_patched_reference(golden, ref_masks, dim, rows):
    """G6: N != 1280 via the reference source with dim/pixscale patched in memory."""
    g = golden('g6_grids')
    lb = g['n%d_lbda' % dim]
    for k in range(rows):
        s, gl, l0, npl, three, ps = g['n%d_r%d_in' % (dim, k)]
        npl, three = int(npl), bool(three)
        tabs = O.ao_tables(H, three, npl, masks=ref_masks)
        fit, fin = O.compute_psf(lb, s, gl, l0, npl, H, three, dim=dim, pixscale=ps, tables=tabs,
                                 fit=(dim == 512))
        assert rel_err(fin, g['n%d_r%d_fin' % (dim, k)]) < 1e-12
        if dim == 512:
            np.testing.assert_allclose(fit[:, 3:], g['n%d_r%d_fit' % (dim, k)][:, 3:], rtol=2e-7)


def test_grid_too_small_raises():
    """psfrec.py:663-683: npixc(lambda) > dim makes the reference raise ValueError."""
    psd = np.ones((128, 128))
    with pytest.raises(ValueError):
        O.psf_stamps_refshaped(psd, np.array([465.0]), 40, 0.2)


def test_convolve_same_is_scipy_fftconvolve():
    from scipy.signal import fftconvolve
    rng = np.random.default_rng(1)
    img = rng.random((40, 40))
    ker = O.moffat_kernel(3.3, 2.2, 41)
    assert rel_err(O.convolve_same(img, ker), fftconvolve(img, ker, mode='same')) < 1e-13


def test_bilinear_sample_is_scipy_interpn():
    from scipy.interpolate import interpn
    rng = np.random.default_rng(2)
    psf = rng.random((254, 254))
    pos = np.mgrid[:40, :40] * 254 / 40
    xin = np.arange(254)
    want = interpn((xin, xin), psf, pos.T, method='linear').T     # psfrec.py:641
    assert rel_err(O._bilinear_sample(psf, 40), want) < 1e-14


def test_sparta_golden_is_self_consistent(golden):
    g = golden('g5_sparta18')
    assert g['fit_rows'].shape == (18, 35, 5)
    fit = O.fit_psf_cube(g['psf_mean'][[0, 17, 34]])
    np.testing.assert_allclose(fit, g['fit_mean'][[0, 17, 34]], rtol=1e-7)
    np.testing.assert_allclose(O.fit_psf_cube(g['fin_row0'][[5]]), g['fit_rows'][0][[5]], rtol=1e-7)


def test_fit_error_columns_follow_the_covariance_recipe(golden):
    """The err_* / flux columns psfrec.py:866-870 keeps from mpdaf's fit object: the oracle's restatement
    (MINPACK cov_x * chi2 / dof) against a covariance built here from a finite-difference Jacobian at the
    solution, and against the FIT_MEAN rows the reference wrote into G7 through it."""
    g = golden('g7_sparta_lgs')
    for tag in ('mean', 'lgs'):
        pm = g[tag + '_psf_mean']
        for k in (0, 3):
            f = O.moffat_fit(pm[k], 0.2, errors=True)
            P, Q = (a.ravel().astype(float) for a in np.indices(pm[k].shape))
            v = np.array([f['peak'], f['center'][0], f['center'][1], f['alpha'], f['n']])
            J = np.empty((P.size, 5))
            for i in range(5):
                h = 1e-6 * max(abs(v[i]), 1.0)
                vp, vm = v.copy(), v.copy()
                vp[i] += h
                vm[i] -= h
                J[:, i] = (O.moffat_model(vp, P, Q) - O.moffat_model(vm, P, Q)) / (2 * h)
            cov = np.linalg.inv(J.T @ J) * f['chi2'] / f['dof']
            err = np.sqrt(np.diag(cov))
            np.testing.assert_allclose([f['err_peak'], f['err_center'][0], f['err_center'][1], f['err_alpha'],
                                        f['err_n']], err, rtol=1e-5)
            # the FWHM's error = what a fit in (fwhm, n) reports: transform the Jacobian instead of the
            # covariance (column of a at fixed fwhm) and invert again
            s2 = 2 ** (1 / f['n']) - 1
            dfw = np.array([2 * np.sqrt(s2), -f['alpha'] * 2 ** (1 / f['n']) * np.log(2) / (f['n'] ** 2 * np.sqrt(s2))])
            Jw = J.copy()
            Jw[:, 3] = J[:, 3] / dfw[0]                       # d/dfw at fixed n
            Jw[:, 4] = J[:, 4] - J[:, 3] * dfw[1] / dfw[0]    # d/dn at fixed fw
            covw = np.linalg.inv(Jw.T @ Jw) * f['chi2'] / f['dof']
            np.testing.assert_allclose(f['err_fwhm'], np.sqrt(covw[3, 3]) * 0.2, rtol=1e-5)
            np.testing.assert_allclose(f['flux'], f['peak'] * np.pi * f['alpha'] ** 2 / (f['n'] - 1), rtol=1e-14)
            # ... and the reference's FIT_MEAN table (psfrec.py:1105) carries exactly these
            for c, want in (('peak', f['peak']), ('flux', f['flux']), ('err_peak', f['err_peak']),
                            ('err_n', f['err_n']), ('err_flux', f['err_flux'])):
                np.testing.assert_allclose(g['%s_mean_%s' % (tag, c)][k], want, rtol=1e-6, err_msg=c)
            np.testing.assert_allclose(g[tag + '_mean_err_fwhm'][k], [f['err_fwhm']] * 2, rtol=1e-6)
            np.testing.assert_allclose(g[tag + '_mean_err_center'][k], f['err_center'], rtol=1e-6)


def test_oracle_reproduces_the_reference_sparta_front_end(golden, ref_masks):
    """G7: the reference's own compute_psf_from_sparta (psfrec.py:981-1120, jittered LGS columns,
    mean_of_lgs True / False) -- the oracle's compute_psf on the task values the reference
    derived, for the 3-LGS row of each mode."""
    g = golden('g7_sparta_lgs')
    lb = np.linspace(float(g['lmin']), float(g['lmax']), int(g['nl']))
    for tag in ('mean', 'lgs'):
        rows = g[tag + '_rows_row_idx']
        three_task = 2 if tag == 'mean' else 5            # row 2 lost a laser (psfrec.py:1053-1054)
        sel = rows == three_task
        see, gl, l0 = g[tag + '_rows_SEEING'][sel][0], g[tag + '_rows_GL'][sel][0], g[tag + '_rows_L0'][sel][0]
        tabs = O.ao_tables(H, True, 1, masks=ref_masks)
        fit, _ = O.compute_psf(lb, see, gl, l0, 1, H, True, tables=tabs)
        np.testing.assert_allclose(fit[:, 3], g[tag + '_rows_fwhm'][sel][:, 0], atol=1e-9)
        np.testing.assert_allclose(fit[:, 4], g[tag + '_rows_n'][sel], atol=1e-8)
    # bookkeeping of the reference: one task per row / per valid laser, 1-based task index
    assert g['mean_rows_lgs_idx'].tolist() == [-1] * 16
    assert g['lgs_rows_lgs_idx'][::4].tolist() == [1, 2, 3, 4, 1, 2, 4, 1, 2, 3, 4, 1, 2, 3, 4]
    assert g['lgs_rows_row_idx'][::4].tolist() == list(range(1, 16))
