"""GPU: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same
inputs, against the golden vectors captured from the real reference, and -- at BASELINE.json's
full sizes -- through size-independent properties.

Tolerances (relative to the stamp maximum for stamps, absolute for fit parameters):
  f64 mode   : stamps 1e-9,  fwhm/beta 1e-6
  mixed mode : stamps 2e-5,  fwhm/beta 1e-4   (north_star: Moffat (fwhm, beta) within 1e-4)
"""
import numpy as np
import pytest

import psfr_oracle as O
from conftest import record_margin,  H, rel_err

pytestmark = pytest.mark.gpu

TOL = {'f64': dict(stamp=1e-9, fit=1e-6), 'mixed': dict(stamp=2e-5, fit=1e-4)}


@pytest.fixture(scope='module')
def api():
    import muse_psfr_amd
    return muse_psfr_amd


def _oracle_tables(npl, masks=None):
    if masks is None:
        return {g: O.ao_tables(H, bool(g), npl, exact_masks=True) for g in (0, 1)}
    return {g: O.ao_tables(H, bool(g), npl, masks=masks) for g in (0, 1)}


@pytest.mark.parametrize('dim,npl', [(128, 1), (256, 3), (512, 1)])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_every_stage_against_the_oracle(api, dim, npl, prec):
    ps = api.grid_pixscale(dim)
    lb = np.array([465.0, 600.0, 930.0])
    cases = [(1.0, 0.7, 25.0, 0), (1.5, 0.3, 10.0, 1), (0.45, 0.93, 28.5, 0)]
    see, gl, l0, three = (np.array([c[k] for c in cases]) for k in range(4))
    ctx = api.Context(dim=dim, pixscale=ps, precision=prec)
    r = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
    ndir = npl * npl
    tab = ctx.debug_fetch('ao_tables', (2, ndir, 3, 80, 80))
    tel = ctx.debug_fetch('tel', (dim // 2 + 1, dim))
    d0 = ctx.debug_fetch('dphi0', (len(cases), ndir, dim // 2 + 1, dim))
    pre = ctx.debug_fetch('pre', (len(cases), lb.size, 40, 40))
    ctx.close()
    tabs = _oracle_tables(npl)
    eps = 1e-12 if prec == 'f64' else 2e-7
    for g in (0, 1):
        T, noise = tabs[g]
        T = T.copy()
        T[..., 0, 0] = 0                                  # psfrec.py:490
        ot = np.stack([np.swapaxes(T[0], -1, -2), np.swapaxes(T[1], -1, -2),
                       np.swapaxes(noise, -1, -2)], axis=1)
        assert rel_err(tab[g], ot) < 1e-12
    otel = O.telescope_otf(dim) * dim * dim
    assert rel_err(tel, otel[:, :dim // 2 + 1].T) < eps
    for k, (s, g_, l, th) in enumerate(cases):
        psd = O.residual_psd([g_, 1 - g_], H, s, l, npl, dim, bool(th), tables=tabs[th])
        od0 = np.array([O.structure_function0(p) for p in psd])
        od0 = np.swapaxes(od0, -1, -2)[:, :dim // 2 + 1, :]
        # (the series form of stage A neither evaluates nor stores the structure function where the telescope
        # OTF vanishes, psfrec.py:784-797: there the buffer keeps the zero it was allocated with)
        inside = np.broadcast_to(tel > 0, d0[k].shape)
        assert np.abs(d0[k] - od0)[inside].max() / np.abs(od0).max() < eps
        out = d0[k][~inside]          # (whole pieces of a line outside the support are skipped, the others computed)
        assert np.all((out == 0) | (np.abs(out - od0[~inside]) / np.abs(od0).max() < eps))
        opre = O.psf_stamps_refshaped(psd, lb, 40, ps)
        assert rel_err(pre[k], opre) < TOL[prec]['stamp']
        ofin = O.convolve_final_psf(lb, s, g_, l, opre, ps)
        assert rel_err(r['psf'][k], ofin) < TOL[prec]['stamp']
        record_margin('every_stage_vs_oracle_%s' % prec, stamp_pre=rel_err(pre[k], opre), stamp=rel_err(r['psf'][k], ofin))
        if dim == 512:           # smaller grids: the stamp is narrower than the PSF, fit ill-posed
            ofit = O.fit_psf_cube(ofin, ps)
            record_margin('every_stage_vs_oracle_%s' % prec,
                          fwhm_arcsec=np.abs(r['fit'][k][:, 5] * ps - ofit[:, 3]).max(),
                          beta=np.abs(r['fit'][k][:, 4] - ofit[:, 4]).max())
            assert np.abs(r['fit'][k][:, 5] * ps - ofit[:, 3]).max() < TOL[prec]['fit']
            assert np.abs(r['fit'][k][:, 4] - ofit[:, 4]).max() < TOL[prec]['fit']
            assert np.abs(r['fit'][k][:, 1:3] - ofit[:, 1:3]).max() < 1e-4
            assert np.all(r['fit'][k][:, 14] == 0)
    np.testing.assert_allclose(r['psf_sum'], r['psf'].sum(axis=0), rtol=1e-13)


@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_native_grid_against_reference_goldens(api, golden, ref_masks, prec):
    """N = 1280, pixscale 0.2: the configuration compute_psf hard-codes (psfrec.py:954-955)."""
    g = golden('g2_native1280')
    lb = g['lbda']
    ctx = api.Context(dim=1280, pixscale=0.2, precision=prec)
    for npl in (1, 3):
        runs = [k for k in range(len(g['meta'])) if int(g['meta'][k][3]) == npl]
        meta = g['meta'][runs]
        r = ctx.reconstruct(lb, meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 4].astype(np.uint8), H,
                            npsflin=npl, masks=ref_masks)
        pre = ctx.debug_fetch('pre', (len(runs), lb.size, 40, 40))
        for i, k in enumerate(runs):
            assert rel_err(pre[i], g['pre_%d' % k]) < TOL[prec]['stamp']
            assert rel_err(r['psf'][i], g['fin_%d' % k]) < TOL[prec]['stamp']
            fit = g['fit_%d' % k]
            record_margin('native1280_reference_goldens_%s' % prec, stamp_pre=rel_err(pre[i], g['pre_%d' % k]),
                          stamp=rel_err(r['psf'][i], g['fin_%d' % k]),
                          fwhm_arcsec=np.abs(r['fit'][i][:, 5] * 0.2 - fit[:, 3]).max(),
                          beta=np.abs(r['fit'][i][:, 4] - fit[:, 4]).max())
            assert np.abs(r['fit'][i][:, 5] * 0.2 - fit[:, 3]).max() < TOL[prec]['fit']
            assert np.abs(r['fit'][i][:, 4] - fit[:, 4]).max() < TOL[prec]['fit']
            assert np.abs(r['fit'][i][:, 0] / fit[:, 0] - 1).max() < TOL[prec]['fit']
    ctx.close()


def test_reference_known_answers_through_the_gpu(api, ref_masks):
    """test_psfrec.py:121-127 (the reference's CLI table) reproduced by the HIP path."""
    ctx = api.Context(dim=1280, pixscale=0.2, precision='mixed')
    r = ctx.reconstruct(np.array([500.0, 700.0, 900.0]), [1.0], [0.7], [25.0], [0], H,
                        masks=ref_masks)
    ctx.close()
    f = r['fit'][0]
    assert ['%.2f' % v for v in f[:, 5] * 0.2] == ['0.85', '0.73', '0.62']
    assert ['%.2f' % v for v in f[:, 4]] == ['2.73', '2.55', '2.23']
    np.testing.assert_allclose(f[:, 1:3], 20, atol=1e-3)


@pytest.mark.parametrize('prec', ['mixed', 'f64'])
@pytest.mark.parametrize('dim', [128, 256, 512, 1024])
def test_patched_grid_goldens(api, golden, ref_masks, dim, prec):
    """G6: the reference source with its hard-coded dim / pixscale patched in memory, for every
    grid of BASELINE.json (incl. the npsflin = 3 row at 256^2 and a 3-LGS row).  On 128^2 / 256^2
    the stamp is narrower than the PSF core and the Moffat fit is ill-posed:
    those grids are graded on the stamps, before and after the convolutions."""
    g = golden('g6_grids')
    lb = g['n%d_lbda' % dim]
    ps = api.grid_pixscale(dim)
    ctx = api.Context(dim=dim, pixscale=ps, precision=prec)
    k = 0
    seen_npl = set()
    while 'n%d_r%d_in' % (dim, k) in g:
        s, gl, l0, npl, three, ps_g = g['n%d_r%d_in' % (dim, k)]
        assert abs(ps_g - ps) < 1e-15
        r = ctx.reconstruct(lb, [s], [gl], [l0], [int(three)], H, npsflin=int(npl), masks=ref_masks)
        pre = ctx.debug_fetch('pre', (1, lb.size, 40, 40))
        assert rel_err(pre[0], g['n%d_r%d_pre' % (dim, k)]) < TOL[prec]['stamp']
        assert rel_err(r['psf'][0], g['n%d_r%d_fin' % (dim, k)]) < TOL[prec]['stamp']
        record_margin('patched_grid_reference_goldens_%d_%s' % (dim, prec),
                      stamp_pre=rel_err(pre[0], g['n%d_r%d_pre' % (dim, k)]),
                      stamp=rel_err(r['psf'][0], g['n%d_r%d_fin' % (dim, k)]))
        if dim >= 512:
            fit = g['n%d_r%d_fit' % (dim, k)]
            record_margin('patched_grid_reference_goldens_%d_%s' % (dim, prec),
                          fwhm_arcsec=np.abs(r['fit'][0][:, 5] * ps - fit[:, 3]).max(),
                          beta=np.abs(r['fit'][0][:, 4] - fit[:, 4]).max())
            assert np.abs(r['fit'][0][:, 5] * ps - fit[:, 3]).max() < TOL[prec]['fit']
            assert np.abs(r['fit'][0][:, 4] - fit[:, 4]).max() < TOL[prec]['fit']
        seen_npl.add(int(npl))
        k += 1
    assert k >= 2 and (dim != 256 or 3 in seen_npl)
    ctx.close()


def test_sparta_table_goldens(api, golden, ref_masks):
    """G5: 18 tasks (two in 3-LGS mode) x 35 wavelengths on the native grid: FIT_ROWS, PSF_MEAN,
    FIT_MEAN semantics of compute_psf_from_sparta (psfrec.py:1086-1113)."""
    g = golden('g5_sparta18')
    ctx = api.Context(dim=1280, pixscale=0.2, precision='mixed')
    ctx.set_option('chunk_tasks', 7)                     # ragged chunks: 7 + 7 + 4
    r = ctx.reconstruct(g['lbda'], g['seeing'], g['gl'], g['l0'], g['three'].astype(np.uint8), H,
                        masks=ref_masks)
    assert np.abs(r['fit'][:, :, 5] * 0.2 - g['fit_rows'][:, :, 3]).max() < 1e-4
    assert np.abs(r['fit'][:, :, 4] - g['fit_rows'][:, :, 4]).max() < 1e-4
    mean = r['psf_sum'] / 18
    assert rel_err(mean, g['psf_mean']) < 1e-5
    fm = ctx.fit_stamps(mean)
    assert np.abs(fm[:, 5] * 0.2 - g['fit_mean'][:, 3]).max() < 1e-4
    assert np.abs(fm[:, 4] - g['fit_mean'][:, 4]).max() < 1e-4
    assert rel_err(r['psf'][0], g['fin_row0']) < 2e-5
    assert rel_err(r['psf'][17], g['fin_row17']) < 2e-5
    ctx.close()


def test_fit_kernel_against_minpack_on_golden_stamps(api, golden):
    """The LM kernel alone: reference-produced stamps in, scipy.optimize.leastsq answers out."""
    g = golden('g2_native1280')
    ctx = api.Context(dim=128, pixscale=0.2, precision='f64')
    for k in range(len(g['meta'])):
        f = ctx.fit_stamps(g['fin_%d' % k])
        want = g['fit_%d' % k]
        np.testing.assert_allclose(f[:, 0], want[:, 0], rtol=1e-7)
        np.testing.assert_allclose(f[:, 1:3], want[:, 1:3], atol=1e-6)
        np.testing.assert_allclose(f[:, 5] * 0.2, want[:, 3], atol=1e-7)
        np.testing.assert_allclose(f[:, 4], want[:, 4], atol=1e-6)
        assert np.all(f[:, 14] == 0)
    ctx.close()


@pytest.mark.parametrize('prec', ['mixed', 'f64'])
def test_fit_error_and_flux_columns_against_the_oracle(api, golden, prec):
    """fit_out[8..13] and [15] (err_peak, err_center, err_alpha, err_n, err_fwhm, flux: the columns
    psfrec.py:866-870 keeps from mpdaf's fit object) and the FIT_ROWS columns the host derives from them,
    on the reference's own final stamps (G2), against the oracle's restatement of mpdaf's recipe
    (MINPACK cov_x * chi2 / dof).  The library takes the covariance from the float normal matrix of the
    last LM iteration, a point within ~1e-4 of the minimum: relative 1e-3."""
    from muse_psfr_amd.psfrec import _fit_columns
    g = golden('g2_native1280')
    ctx = api.Context(dim=128, pixscale=0.2, precision=prec)
    worst = {}
    for k in range(len(g['meta'])):
        fin = g['fin_%d' % k]
        f = ctx.fit_stamps(fin)
        cols = _fit_columns(np.arange(len(fin), dtype=float), f, 0.2)
        for j in range(len(fin)):
            o = O.moffat_fit(fin[j], 0.2, errors=True)
            got = dict(err_peak=f[j, 8], err_center0=f[j, 9], err_center1=f[j, 10], err_alpha=f[j, 11],
                       err_n=f[j, 12], err_fwhm=f[j, 13] * 0.2, flux=f[j, 15], chi2=f[j, 6],
                       col_err_flux=cols['err_flux'][j], col_err_fwhm=cols['err_fwhm'][j, 1],
                       col_err_center=cols['err_center'][j, 1], col_err_n=cols['err_n'][j],
                       col_err_peak=cols['err_peak'][j], col_flux=cols['flux'][j])
            want = dict(err_peak=o['err_peak'], err_center0=o['err_center'][0], err_center1=o['err_center'][1],
                        err_alpha=o['err_alpha'], err_n=o['err_n'], err_fwhm=o['err_fwhm'], flux=o['flux'],
                        chi2=o['chi2'], col_err_flux=o['err_flux'], col_err_fwhm=o['err_fwhm'],
                        col_err_center=o['err_center'][1], col_err_n=o['err_n'], col_err_peak=o['err_peak'],
                        col_flux=o['flux'])
            for c in got:
                worst[c] = max(worst.get(c, 0.0), abs(got[c] / want[c] - 1))
    ctx.close()
    record_margin('fit_error_columns_%s' % prec, **{'rel_' + c: v for c, v in worst.items()})
    assert worst['flux'] < 1e-5 and worst['col_flux'] < 1e-5, worst
    assert max(worst.values()) < 1e-3, worst


def test_chunking_and_repeat_are_bitwise_invariant(api):
    see, gl, l0 = api.synthetic_rows(9)
    lb = np.linspace(465, 930, 5)
    ps = api.grid_pixscale(256)
    ctx = api.Context(dim=256, pixscale=ps, precision='mixed')
    three = np.array([0, 1, 0, 0, 1, 0, 0, 0, 1], np.uint8)
    a = ctx.reconstruct(lb, see, gl, l0, three, H)
    b = ctx.reconstruct(lb, see, gl, l0, three, H)
    ctx.set_option('chunk_tasks', 2)
    c = ctx.reconstruct(lb, see, gl, l0, three, H)
    ctx.close()
    assert np.array_equal(a['psf'], b['psf']) and np.array_equal(a['fit'], b['fit'])
    assert np.array_equal(a['psf'], c['psf']) and np.array_equal(a['fit'], c['fit'])
    np.testing.assert_allclose(a['psf_sum'], c['psf_sum'], rtol=1e-14)
    # row order equivariance
    perm = np.array([3, 0, 8, 1, 7, 2, 6, 4, 5])
    ctx = api.Context(dim=256, pixscale=ps, precision='mixed')
    d = ctx.reconstruct(lb, see[perm], gl[perm], l0[perm], three[perm], H)
    ctx.close()
    assert np.array_equal(d['psf'], a['psf'][perm])


def test_wide_parameter_range_against_the_oracle(api):
    """Rows drawn over the whole validity window of the SPARTA filter (psfrec.py:1049-1051) and
    beyond the bench distribution: seeing 0.3-2.5 arcsec, GL 0.02-0.98, L0 8.1-29.9 m."""
    rng = np.random.default_rng(2024)
    n = 20
    see = rng.uniform(0.3, 2.5, n)
    gl = rng.uniform(0.02, 0.98, n)
    l0 = rng.uniform(8.1, 29.9, n)
    three = (rng.random(n) < 0.3).astype(np.uint8)
    see[:2], gl[:2], l0[:2] = [0.3, 2.5], [0.98, 0.02], [29.9, 8.1]      # corners
    lb = np.array([465.0, 640.0, 930.0])
    ps = api.grid_pixscale(512)
    ctx = api.Context(dim=512, pixscale=ps, precision='mixed')
    r = ctx.reconstruct(lb, see, gl, l0, three, H)
    ctx.close()
    tabs = _oracle_tables(1)
    worst = [0.0, 0.0, 0.0]
    nwell = 0
    for k in range(n):
        _, ofin = O.compute_psf(lb, see[k], gl[k], l0[k], 1, H, bool(three[k]), dim=512,
                                pixscale=ps, tables=tabs[int(three[k])], fit=False)
        worst[0] = max(worst[0], rel_err(r['psf'][k], ofin))
        for j in range(lb.size):
            (pk, p0, q0, fw, beta), chi2, _ = O.moffat_fit(ofin[j], ps, full=True)
            g = r['fit'][k][j]
            if beta < 10:      # well-posed: the stamp is wider than the PSF core
                nwell += 1
                worst[1] = max(worst[1], abs(g[5] * ps - fw))
                worst[2] = max(worst[2], abs(g[4] - beta))
                assert abs(g[6] - chi2) <= 1e-3 * chi2, (k, j, g[6], chi2)   # float-evaluated chi2
            else:
                # very broad PSFs (seeing > 2 arcsec on this 3 arcsec stamp) drive beta -> 1e3..1e4
                # along a flat valley in both solvers (MINPACK crawls further with xtol = 1e-14):
                # only the objective is comparable, loosely
                assert np.isfinite(g).all() and g[6] <= 1.5 * chi2, (k, j, g[6], chi2)
    assert nwell >= 40
    record_margin('wide_parameter_range_vs_oracle', stamp=worst[0], fwhm_arcsec=worst[1], beta=worst[2])
    assert worst[0] < 2e-5 and worst[1] < 1e-4 and worst[2] < 1e-4, worst


def test_pipeline_lanes_and_kernel_variants_agree(api):
    """Two HIP-stream lanes vs one; FFT vs direct convolution; hardware exp vs expf."""
    see, gl, l0 = api.synthetic_rows(64)
    lb = np.linspace(465, 930, 35)
    ps = api.grid_pixscale(128)
    three = (np.arange(64) % 5 == 0).astype(np.uint8)
    res = {}
    for key, opts in (('base', {'streams': 2}), ('one_lane', {'streams': 1}), ('auto', {}),
                      ('direct_conv', {'fft_conv': 0}),
                      ('expf', {'fast_exp': 0}),
                      ('cu_masks', {'streams': 2, 'cu_partition': 1}),       # lanes on CU-masked streams
                      ('memcpy_params', {'param_copy': 0})):                 # the parameter blob by hipMemcpyAsync
        ctx = api.Context(dim=128, pixscale=ps, precision='mixed')
        for k, v in opts.items():
            ctx.set_option(k, v)
        res[key] = ctx.reconstruct(lb, see, gl, l0, three, H)      # 2240 stamps: split over 2 lanes
        if key == 'cu_masks':
            ctx.set_option('cu_partition', 0)                      # (the lanes' streams are created anew)
            again = ctx.reconstruct(lb, see, gl, l0, three, H)
            assert np.array_equal(again['fit'], res[key]['fit'])
        ctx.close()
    a, b = res['base'], res['one_lane']
    assert np.array_equal(a['psf'], b['psf']) and np.array_equal(a['fit'], b['fit'])
    for key in ('cu_masks', 'memcpy_params'):                      # where and how the work is queued changes no bit
        assert np.array_equal(res[key]['psf'], a['psf']) and np.array_equal(res[key]['fit'], a['fit']), key
    assert np.array_equal(res['auto']['fit'], b['fit'])      # one chunk: automatic = one lane
    np.testing.assert_allclose(a['psf_sum'], b['psf_sum'], rtol=1e-13)
    for key in ('direct_conv', 'expf'):
        assert rel_err(res[key]['psf'], a['psf']) < 1e-5, key
    with pytest.raises(api.MpsfrError):
        api.Context(dim=128, pixscale=ps).set_option('streams', 5)


@pytest.mark.parametrize('streams', [3, 4])
@pytest.mark.parametrize('ntask,chunk', [(10, 5), (12, 5), (7, 7)])
def test_more_lanes_than_chunks(api, streams, ntask, chunk):
    """A call with fewer chunks than pipeline lanes (2 or 3 chunks, or 1, on 3-4 lanes): lanes
    without a chunk must not contribute to the stamp sum.  With at most one chunk per lane the sum
    over lanes adds the chunk sums in the same order as a single lane does: bit for bit."""
    see, gl, l0 = api.synthetic_rows(ntask)
    lb = np.linspace(465, 930, 4)
    ps = api.grid_pixscale(128)
    three = (np.arange(ntask) % 4 == 1).astype(np.uint8)
    out = {}
    for key, st in (('one', 1), ('many', streams)):
        ctx = api.Context(dim=128, pixscale=ps, precision='mixed')
        ctx.set_option('streams', st)
        ctx.set_option('chunk_tasks', chunk)
        # dirty the per-lane partial sums with a call that uses every lane
        if st > 1:
            ctx.reconstruct(lb, *api.synthetic_rows(4 * st, seed=9), np.zeros(4 * st, np.uint8), H)
            ctx.set_option('chunk_tasks', chunk)
        out[key] = ctx.reconstruct(lb, see, gl, l0, three, H)
        ctx.close()
    a, b = out['one'], out['many']
    assert np.array_equal(a['psf'], b['psf']) and np.array_equal(a['fit'], b['fit'])
    np.testing.assert_array_equal(a['psf_sum'], b['psf_sum'])
    np.testing.assert_allclose(b['psf_sum'], b['psf'].sum(axis=0), rtol=1e-13)


def test_edge_cases_and_errors(api):
    from muse_psfr_amd import MpsfrError
    ps = api.grid_pixscale(128)
    ctx = api.Context(dim=128, pixscale=ps, precision='mixed')
    r = ctx.reconstruct([700.0], [1.0], [0.7], [25.0], [0], H)           # 1 task, 1 wavelength
    assert r['psf'].shape == (1, 1, 40, 40) and np.isfinite(r['psf']).all()
    r2 = ctx.reconstruct([700.0], [1.0], [0.7], [25.0], [0], H, want_psf=False, want_fit=False)
    assert r2['psf'] is None and r2['fit'] is None
    np.testing.assert_array_equal(r2['psf_sum'][0], r['psf'][0, 0])
    with pytest.raises(MpsfrError) as e:                                 # psfrec.py:663-683
        ctx.reconstruct([300.0], [1.0], [0.7], [25.0], [0], H)
    assert e.value.code == -3
    with pytest.raises(MpsfrError):
        ctx.reconstruct([700.0], [-1.0], [0.7], [25.0], [0], H)
    with pytest.raises(MpsfrError):
        ctx.reconstruct([700.0], [1.0], [0.7], [25.0], [0], H, npsflin=9)
    ctx.close()
    with pytest.raises(MpsfrError):
        api.Context(dim=500, pixscale=0.2)
    with pytest.raises(MpsfrError):
        api.Context(dim=512, pixscale=0.2, dimpsf=64)
    # reference grid with the reference's pixel scale: 465 nm does not fit (npixc = 1336 > 1280)
    ctx = api.Context(dim=1280, pixscale=0.2)
    with pytest.raises(MpsfrError) as e:
        ctx.reconstruct([465.0], [1.0], [0.7], [25.0], [0], H)
    assert e.value.code == -3
    ctx.close()


def test_profile_options(api):
    """Per-kernel HIP-event timing: all kernels, or only the one named by profile_only; and
    back-to-back calls without a host wait in between (ring of parameter blobs)."""
    see, gl, l0 = api.synthetic_rows(6)
    lb = np.linspace(500, 900, 4)
    ctx = api.Context(dim=128, pixscale=api.grid_pixscale(128))
    names = ctx.profile_names()
    assert 'otf_mfma' in names and 'otf_rowfft' in names and 'fit' in names
    ctx.set_option('profile', 1)
    ctx.profile_reset()
    first = ctx.reconstruct(lb, see, gl, l0, np.zeros(6, np.uint8), H)
    prof = ctx.profile()
    assert prof['otf_mfma'][1] == 1 and prof['fit'][1] == 1 and prof['otf_mfma'][0] > 0
    assert prof['otf_rowfft'][1] == 0 and prof['mf_prep'][1] == 1       # block minima + masks
    ctx.set_option('profile_only', names.index('otf_mfma'))
    ctx.profile_reset()
    import torch
    from muse_psfr_amd import NFIT
    dev = torch.device('cuda:0')
    fit = torch.zeros((6, 4, NFIT), dtype=torch.float64, device=dev)
    psum = torch.zeros((4, 40, 40), dtype=torch.float64, device=dev)
    for _ in range(9):          # more calls in flight than parameter blobs
        ctx.reconstruct_device(lb, see, gl, l0, np.zeros(6, np.uint8), H, 12.0, 1, None, None,
                               psum.data_ptr(), fit.data_ptr())
    ctx.sync()
    prof = ctx.profile()
    assert prof['otf_mfma'][1] == 9 and prof['fit'][1] == 0
    np.testing.assert_array_equal(fit.cpu().numpy(), first['fit'])
    with pytest.raises(api.MpsfrError):
        ctx.set_option('profile_only', 99)
    ctx.close()


def test_context_pool_pipelines_independent_batches(api):
    """Two contexts fed in turn: same results as one context, batch by batch."""
    import torch
    from muse_psfr_amd import NFIT
    lb = np.linspace(480, 920, 5)
    dev = torch.device('cuda:0')
    ps = api.grid_pixscale(128)
    batches = [api.synthetic_rows(7, seed=100 + k) for k in range(6)]
    one = api.Context(dim=128, pixscale=ps)
    ref = [one.reconstruct(lb, *b, np.zeros(7, np.uint8), H)['fit'] for b in batches]
    one.close()
    with api.ContextPool(2, dim=128, pixscale=ps) as pool:
        fits = [torch.zeros((7, 5, NFIT), dtype=torch.float64, device=dev) for _ in batches]
        psum = [torch.zeros((5, 40, 40), dtype=torch.float64, device=dev) for _ in batches]
        for k, b in enumerate(batches):
            pool.next().reconstruct_device(lb, *b, np.zeros(7, np.uint8), H, 12.0, 1, None, None,
                                           psum[k].data_ptr(), fits[k].data_ptr())
        pool.sync()
        for k in range(len(batches)):
            np.testing.assert_array_equal(fits[k].cpu().numpy(), ref[k])


def test_float_altitudes_use_wind_12p5(api):
    """psfrec.py:61: np.full_like(h, 12.5) truncates to 12 only for integer altitudes."""
    ps = api.grid_pixscale(256)
    lb = np.array([500.0, 900.0])
    ctx = api.Context(dim=256, pixscale=ps, precision='f64')
    r = ctx.reconstruct(lb, [1.0], [0.7], [25.0], [0], (100.0, 10000.0))
    ctx.close()
    tabs = O.ao_tables((100.0, 10000.0), False, 1, exact_masks=True)
    _, fin = O.compute_psf(lb, 1.0, 0.7, 25.0, 1, (100.0, 10000.0), False, dim=256, pixscale=ps,
                           tables=tabs, fit=False)
    assert rel_err(r['psf'][0], fin) < 1e-9
    tabs12 = O.ao_tables((100, 10000), False, 1, exact_masks=True)
    _, fin12 = O.compute_psf(lb, 1.0, 0.7, 25.0, 1, (100, 10000), False, dim=256, pixscale=ps,
                             tables=tabs12, fit=False)
    assert rel_err(fin12, fin) > 1e-7        # the two wind speeds are distinguishable


def test_device_pointer_path_matches_host_path(api):
    import torch
    from muse_psfr_amd import NFIT
    see, gl, l0 = api.synthetic_rows(5)
    lb = np.linspace(465, 930, 4)
    ps = api.grid_pixscale(256)
    ctx = api.Context(dim=256, pixscale=ps, precision='mixed')
    a = ctx.reconstruct(lb, see, gl, l0, np.zeros(5, np.uint8), H)
    dev = torch.device('cuda', 0)
    psf = torch.zeros((5, 4, 40, 40), dtype=torch.float64, device=dev)
    psum = torch.zeros((4, 40, 40), dtype=torch.float64, device=dev)
    fit = torch.zeros((5, 4, NFIT), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx.reconstruct_device(lb, see, gl, l0, np.zeros(5, np.uint8), H, 12.0, 1, None,
                           psf.data_ptr(), psum.data_ptr(), fit.data_ptr())
    ctx.sync()
    assert np.array_equal(psf.cpu().numpy(), a['psf'])
    assert np.array_equal(fit.cpu().numpy(), a['fit'])
    assert np.array_equal(psum.cpu().numpy(), a['psf_sum'])
    ctx.close()


def test_full_size_properties(api):
    """BASELINE.json configs[1]: 100 rows x 35 wavelengths on 512^2 -- too large for the oracle
    in a unit test, so size-independent properties are checked instead."""
    see, gl, l0 = api.synthetic_rows(100)
    lb = np.linspace(465, 930, 35)
    ps = api.grid_pixscale(512)
    ctx = api.Context(dim=512, pixscale=ps, precision='mixed')
    r = ctx.reconstruct(lb, see, gl, l0, np.zeros(100, np.uint8), H)
    psf, fit = r['psf'], r['fit']
    assert np.isfinite(psf).all() and np.isfinite(fit).all()
    # checksum of checksums: the deterministic chunked reduction equals the sum of the stamps
    np.testing.assert_allclose(r['psf_sum'], psf.sum(axis=0), rtol=1e-13)
    # zero-padded 'same' convolutions lose a little flux and never gain any (psfrec.py:917, 928)
    flux = psf.sum(axis=(2, 3))
    assert np.all(flux < 1.0) and np.all(flux > 0.9)
    assert np.all(psf >= 0)
    # the peak sits on the stamp centre (test_psfrec.py:28 'center == 20')
    am = psf.reshape(100, 35, -1).argmax(axis=2)
    assert np.all(am == 20 * 40 + 20)
    assert np.abs(fit[:, :, 1:3] - 20).max() < 0.1
    assert np.all(fit[:, :, 14] == 0)
    # FWHM is smaller at 930 nm than at 465 nm for every row; worse seeing gives a wider PSF
    assert np.all(fit[:, -1, 5] < fit[:, 0, 5])
    order = np.argsort(see)
    assert fit[order[-1], 0, 5] > fit[order[0], 0, 5]
    # idempotence of the fit: refitting the returned stamps reproduces the fit table
    f2 = ctx.fit_stamps(psf[:3].reshape(-1, 40, 40))
    np.testing.assert_array_equal(f2, fit[:3].reshape(-1, fit.shape[-1]))
    # the oracle on a sample of rows at full size
    for k in (0, 57):
        tabs = O.ao_tables(H, False, 1, exact_masks=True)
        ofit, ofin = O.compute_psf(lb[[0, 17, 34]], see[k], gl[k], l0[k], 1, H, False, dim=512,
                                   pixscale=ps, tables=tabs)
        assert rel_err(psf[k][[0, 17, 34]], ofin) < 2e-5
        assert np.abs(fit[k][[0, 17, 34], 5] * ps - ofit[:, 3]).max() < 1e-4
        assert np.abs(fit[k][[0, 17, 34], 4] - ofit[:, 4]).max() < 1e-4
    ctx.close()


def test_pipelined_async_calls_match_sequential_ones(api):
    """Asynchronous (on_device) calls queued back to back without a host sync, first into separate
    device buffers, then reusing them, give exactly the results of synchronous calls."""
    import torch
    from muse_psfr_amd import NFIT
    ps = api.grid_pixscale(256)
    lb = np.linspace(465, 930, 35)
    see, gl, l0 = api.synthetic_rows(5 * 60)
    dev = torch.device('cuda', 0)
    ctx = api.Context(dim=256, pixscale=ps, precision='mixed')
    ref = []
    for b in range(5):
        sl = slice(b * 60, (b + 1) * 60)
        ref.append(ctx.reconstruct(lb, see[sl], gl[sl], l0[sl], np.zeros(60, np.uint8), H,
                                   want_psf=False))
    fits = [torch.zeros((60, 35, NFIT), dtype=torch.float64, device=dev) for _ in range(5)]
    sums = [torch.zeros((35, 40, 40), dtype=torch.float64, device=dev) for _ in range(5)]
    torch.cuda.synchronize()
    for rep in range(2):                      # the second round reuses the buffers
        for b in range(5):
            sl = slice(b * 60, (b + 1) * 60)
            ctx.reconstruct_device(lb, see[sl], gl[sl], l0[sl], np.zeros(60, np.uint8), H, 12.0, 1,
                                   None, None, sums[b].data_ptr(), fits[b].data_ptr())
    ctx.sync()
    for b in range(5):
        assert np.array_equal(fits[b].cpu().numpy(), ref[b]['fit']), b
        assert np.array_equal(sums[b].cpu().numpy(), ref[b]['psf_sum']), b
    ctx.close()


def test_consecutive_calls_overlap_on_lanes_and_stay_ordered_per_buffer(api):
    """Asynchronous calls of one context rotate over its lanes: distinct output buffers may run
    concurrently, a reused buffer ends with the LAST call's result, and mpsfr_wait_event holds a
    call back behind the caller's own GPU work."""
    import torch
    from muse_psfr_amd import NFIT
    ps = api.grid_pixscale(128)
    lb = np.linspace(465, 930, 6)
    dev = torch.device('cuda', 0)
    batches = [api.synthetic_rows(9, seed=500 + k) for k in range(5)]
    z = np.zeros(9, np.uint8)
    ref = api.Context(dim=128, pixscale=ps)
    want = [ref.reconstruct(lb, *b, z, H, want_psf=False) for b in batches]
    ref.close()
    ctx = api.Context(dim=128, pixscale=ps)
    fits = [torch.zeros((9, 6, NFIT), dtype=torch.float64, device=dev) for _ in batches]
    sums = [torch.zeros((6, 40, 40), dtype=torch.float64, device=dev) for _ in batches]
    for k, b in enumerate(batches):                       # distinct buffers
        ctx.reconstruct_device(lb, *b, z, H, 12.0, 1, None, None, sums[k].data_ptr(), fits[k].data_ptr())
    ctx.sync()
    for k in range(len(batches)):
        assert np.array_equal(fits[k].cpu().numpy(), want[k]['fit']), k
        assert np.array_equal(sums[k].cpu().numpy(), want[k]['psf_sum']), k
    for k, b in enumerate(batches):                       # one buffer, five calls: the last wins
        ctx.reconstruct_device(lb, *b, z, H, 12.0, 1, None, None, sums[0].data_ptr(), fits[0].data_ptr())
    ctx.sync()
    assert np.array_equal(fits[0].cpu().numpy(), want[-1]['fit'])
    assert np.array_equal(sums[0].cpu().numpy(), want[-1]['psf_sum'])
    # a call held back by a caller's event: the caller's copy of the old content must complete
    # before the call overwrites the buffer
    keep = torch.empty_like(fits[1])
    big = torch.randn(4096, 4096, device=dev)
    for _ in range(4):
        big = big @ big * 1e-4                            # keeps torch's stream busy for a while
    keep.copy_(fits[1])
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    ctx.wait_event(ev.cuda_event)
    ctx.reconstruct_device(lb, *batches[3], z, H, 12.0, 1, None, None, sums[1].data_ptr(), fits[1].data_ptr())
    ctx.sync()
    torch.cuda.synchronize()
    assert np.array_equal(keep.cpu().numpy(), want[1]['fit'])
    assert np.array_equal(fits[1].cpu().numpy(), want[3]['fit'])
    ctx.set_option('pipeline_calls', 0)
    ctx.reconstruct_device(lb, *batches[2], z, H, 12.0, 1, None, None, sums[2].data_ptr(), fits[2].data_ptr())
    ctx.sync()
    assert np.array_equal(fits[2].cpu().numpy(), want[2]['fit'])
    ctx.close()


@pytest.mark.parametrize('prec', ['mixed', 'f64'])
def test_reserved_cus_and_head_fusion_change_no_bit(api, prec):
    """Round 6: where the work of a call is queued changes no bit of its results -- the persistent grids of
    K_OTF_MFMA2 / K_DPHI_SERIES with R CUs left free ("persist_reserve": 0, 32, 100, automatic), the kernel spectra of
    the tip-tilt kernels as workgroups of K_PATCH_ROWS or as a kernel of their own ("head_fusion"), the parameter
    blob fetched by workgroups of K_PATCH_GEN ("copy_fusion"), the stamps K_OTF_MFMA2 split into sweeps finished by
    K_CONV_FFT or by K_MF_FINISH ("finish_fusion") -- for host-output calls (one chunk, several chunks on
    two lanes) and for lean device-output calls queued back to back (series form of stage A: 512^2)."""
    import torch
    from muse_psfr_amd import NFIT
    n = 37
    see, gl, l0 = api.synthetic_rows(n, seed=77)
    three = (np.arange(n) % 4 == 0).astype(np.uint8)
    lb = np.linspace(480, 930, 7)
    ps = api.grid_pixscale(512)
    dev = torch.device('cuda', 0)
    variants = (('base', {'persist_reserve': 0, 'head_fusion': 0}),
                ('reserve32', {'persist_reserve': 32, 'head_fusion': 0}),
                ('reserve100_fused', {'persist_reserve': 100}),
                ('defaults', {}),
                ('copy_fused', {'copy_fusion': 1}),
                ('finish_fused', {'finish_fusion': 1}),           # K_CONV_FFT finishes the split stamps itself
                ('stage_a_queue', {'stage_a_queue': 1}),          # K_DPHI_SERIES_Q: the same lines dealt from a queue
                ('copy_fused_chunks', {'copy_fusion': 1, 'chunk_tasks': 10, 'streams': 2}),
                ('fused_chunks', {'chunk_tasks': 10, 'streams': 2}))
    ref = None
    for key, opts in variants:
        ctx = api.Context(dim=512, pixscale=ps, precision=prec)
        for k, v in opts.items():
            ctx.set_option(k, v)
        for _ in range(2):                                   # (the second call finds every table cached)
            r = ctx.reconstruct(lb, see, gl, l0, three, H)
        # lean calls: device outputs, no host synchronisation in between, buffers reused
        fits = [torch.zeros((n, 7, NFIT), dtype=torch.float64, device=dev) for _ in range(2)]
        sums = [torch.zeros((7, 40, 40), dtype=torch.float64, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        for rep in range(6):
            ctx.reconstruct_device(lb, see, gl, l0, three, H, 12.0, 1, None, None, sums[rep % 2].data_ptr(),
                                   fits[rep % 2].data_ptr())
        ctx.sync()
        if ref is None:
            ref = r
        for name in ('psf', 'fit', 'psf_sum'):
            if name == 'psf_sum' and 'chunk_tasks' in opts:  # (per-lane partial sums: another order of additions)
                np.testing.assert_allclose(r[name], ref[name], rtol=1e-13, err_msg=key)
            else:
                assert np.array_equal(r[name], ref[name]), (key, name)
        if 'chunk_tasks' not in opts:
            for b in range(2):
                assert np.array_equal(fits[b].cpu().numpy(), ref['fit']), (key, b)
                assert np.array_equal(sums[b].cpu().numpy(), ref['psf_sum']), (key, b)
        ctx.close()
    with pytest.raises(api.MpsfrError):
        api.Context(dim=512, pixscale=ps).set_option('persist_reserve', 300)


def test_lines_stage_b_drops_may_be_skipped_in_stage_a(api):
    """`stage_a_queue` = 2 (measured, not the default: profiles/r06_experiments.md): K_DPHI_SERIES_Q skips the lines of a
    task on which a lower bound of the structure function from the patch's row transforms alone -- D(., y) >=
    scale2 (sum P - sum_su |T[y][su]|) -- puts every element of the OTF below the eps rule of the pruning, or the
    line's whole mass below its share of the tier budget, at the longest wavelength.  On rows with poor seeing and a
    weak ground layer some lines go, the stamps stay within 1e-6 of their peak and the fits within 1e-6, the lines
    that are computed are bit-identical, and what was skipped really is where the OTF is negligible: the full
    structure function there is above the bound the rule needs."""
    n, dim = 24, 1280
    see = np.linspace(0.6, 1.6, n)
    gl = np.linspace(0.9, 0.35, n)
    l0 = np.linspace(12.0, 28.0, n)
    lb = np.linspace(490.0, 930.0, 5)
    res = {}
    for mode in (0, 2):
        ctx = api.Context(dim=dim, pixscale=0.2)
        ctx.set_option('stage_a_queue', mode)
        r = ctx.reconstruct(lb, see, gl, l0, None, H)
        d = ctx.debug_fetch('dphi0', (n, 1, dim // 2 + 1, dim))[:, 0]
        tel = ctx.debug_fetch('tel', (dim // 2 + 1, dim))
        ctx.close()
        res[mode] = (r, d)
    (ra, da), (rb, db) = res[0], res[2]
    inside = tel > 0
    skipped = ((db >= 1e29) | ~inside[None]).all(axis=2) & inside.any(axis=1)[None]       # (task, line)
    assert 0.02 < skipped.mean() < 0.9, skipped.mean()
    assert not skipped[:, :4].any()
    kept = ~skipped
    assert np.array_equal(da[kept], db[kept])
    # every element of a skipped line: 2^(c' D) tel below 2^-29 of OTF[0][0] = 1 even at 930 nm
    c2 = -0.5 * (2 * np.pi / 930.0) ** 2 * np.log2(np.e)
    with np.errstate(divide='ignore'):
        e = c2 * da + np.log2(np.where(inside, tel, 0.0))[None]
    assert e[skipped].max() < -29.0, e[skipped].max()
    peak = ra['psf'].max(axis=(2, 3), keepdims=True)
    assert (np.abs(rb['psf'] - ra['psf']) / peak).max() < 1e-6
    well = ra['fit'][..., 14] == 0
    assert np.abs(rb['fit'][..., 4:6] - ra['fit'][..., 4:6])[well].max() < 1e-6


def test_blocks_any_wavelength_keeps_and_the_telescope_support(api):
    """`mf_work[5..6]` (VERDICT r5 #2b): the blocks of the half plane some wavelength of a task keeps -- what stage A
    has to deliver at all -- and the blocks inside the support of the telescope OTF.  Bounds that hold by construction:
    executed tile steps / nl <= union <= support x tasks <= all blocks x tasks; the support is the disc of radius N/2
    (pi / 4 of the half plane, a little more in whole blocks); a task with poor seeing and a weak ground layer keeps a
    small core, one with good seeing everything inside the support."""
    dim, nl = 512, 7
    ps = api.grid_pixscale(dim)
    lb = np.linspace(465.0, 930.0, nl)
    ctx = api.Context(dim=dim, pixscale=ps)
    nblocks = ((dim // 2 + 1 + 15) // 16) * (dim // 32)
    fr = {}
    for key, (see, gl, l0) in (('poor', (1.5, 0.45, 25.0)), ('good', (0.5, 0.9, 10.0))):
        ctx.reconstruct(lb, [see], [gl], [l0], want_psf=False)
        w = ctx.debug_fetch('mf_work', (7,))
        steps, tiles, total, full, mid, union, support = w
        assert total == nl * nblocks and steps == full + mid
        assert steps / nl <= union <= support <= nblocks
        assert 0.78 < support / nblocks < 0.83
        fr[key] = union / nblocks
    ctx.close()
    assert fr['poor'] < 0.1 and fr['good'] > 0.7, fr


def test_stream_wait_hands_results_to_a_caller_stream(api):
    """mpsfr_stream_wait is how `bench.py --gpus N` hands a call's device outputs to the stream of the
    collectives.  Lean calls (device outputs, one lane, the context's stream never asked for) into REUSED buffers;
    the caller's stream waits through stream_wait and copies with no host synchronisation: the copies equal the
    blocking call's results bit for bit -- also with the chunks of a call on two lanes (the join + stream_tail case)."""
    import torch
    from muse_psfr_amd import NFIT
    n, nl = 60, 9
    lb = np.linspace(465, 930, nl)
    ps = api.grid_pixscale(512)
    dev = torch.device('cuda', 0)
    batches = [api.synthetic_rows(n, seed=900 + k) for k in range(4)]
    z = np.zeros(n, np.uint8)
    ref = api.Context(dim=512, pixscale=ps)
    want = [ref.reconstruct(lb, *b, z, H, want_psf=False) for b in batches]
    ref.close()
    for opts in ({}, {'chunk_tasks': 16, 'streams': 2}):
        ctx = api.Context(dim=512, pixscale=ps)
        for k, v in opts.items():
            ctx.set_option(k, v)
        fit = torch.zeros((n, nl, NFIT), dtype=torch.float64, device=dev)
        psum = torch.zeros((nl, 40, 40), dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev)
        got_fit, got_sum = [], []
        torch.cuda.synchronize()
        for k, b in enumerate(batches):
            # the call that overwrites the buffers waits for the copy that last read them
            if k:
                ev = torch.cuda.Event()
                ev.record(side)
                ctx.wait_event(ev.cuda_event)
            ctx.reconstruct_device(lb, *b, z, H, 12.0, 1, None, None, psum.data_ptr(), fit.data_ptr())
            ctx.stream_wait(side.cuda_stream)
            with torch.cuda.stream(side):
                got_fit.append(fit.clone())
                got_sum.append(psum.clone())
        side.synchronize()
        ctx.sync()
        for k in range(len(batches)):
            assert np.array_equal(got_fit[k].cpu().numpy(), want[k]['fit']), (opts, k)
            if opts:
                np.testing.assert_allclose(got_sum[k].cpu().numpy(), want[k]['psf_sum'], rtol=1e-13)
            else:
                assert np.array_equal(got_sum[k].cpu().numpy(), want[k]['psf_sum']), (opts, k)
        ctx.close()


@pytest.mark.parametrize('dim', [128, 256, 512])
def test_ill_conditioned_fits_are_flagged_and_the_others_agree(api, dim):
    """VERDICT r5 #7.  Where the stamp is narrower than the PSF core (the small grids with the rescaled pixel scale:
    FWHM 30 px in a 40 px stamp) the least-squares minimum exists but (fwhm, n) are not pinned to 1e-4 by stamps known
    to a few 1e-7 of their peak.  The fit kernel says so itself: status bit MPSFR_FIT_ILL_CONDITIONED when
    n^2 sqrt(cov[eta, eta]) peak >= 100 (noise of 1e-6 of the peak moves n by 1e-4), from the covariance err_n
    comes from.  On 257 rows x 5 wavelengths: the stamps flagged in neither precision agree between the mixed and the
    f64 mode to 6e-5 in n (the round-5 sweep, with its ad-hoc  beta < 20 & fwhm > 2.5 px, stood at 8.0e-5 on 256^2),
    the worst of them also against the oracle; at 512^2 nothing is flagged (sensitivity < 30) and they agree to 2e-5."""
    from muse_psfr_amd import FIT_ILL_CONDITIONED
    ps = api.grid_pixscale(dim)
    nl, rows = 5, 257
    lb = np.linspace(465.0, 930.0, nl)
    see, gl, l0 = api.synthetic_rows(rows)
    three = (np.arange(rows) % 3 == 1).astype(np.uint8)
    res = {}
    for prec in ('mixed', 'f64'):
        ctx = api.Context(dim=dim, pixscale=ps, precision=prec)
        res[prec] = ctx.reconstruct(lb, see, gl, l0, three, H)
        ctx.close()
    a, b = res['mixed']['fit'], res['f64']['fit']
    assert set(np.unique(a[..., 14])) <= {0.0, float(FIT_ILL_CONDITIONED)}
    sens = b[..., 12] * b[..., 0] / np.sqrt(b[..., 6] / 1595.0)          # err_n peak / sqrt(chi2 / dof)
    flagged = b[..., 14] == FIT_ILL_CONDITIONED
    assert np.all(flagged == (sens >= 100.0 * (1 - 1e-9))) or np.abs(sens[flagged != (sens >= 100)] - 100).max() < 1e-3
    well = (a[..., 14] == 0) & (b[..., 14] == 0)
    dn = np.abs(a[..., 4] - b[..., 4])
    dw = np.abs(a[..., 5] - b[..., 5]) * ps
    k = np.unravel_index(np.argmax(np.where(well, dn, 0)), dn.shape)
    record_margin('ill_conditioned_rule_dim%d' % dim, beta=dn[well].max(), fwhm_arcsec=dw[well].max(),
                  flagged_fraction=flagged.mean(), worst_sensitivity_of_the_unflagged=sens[well].max())
    if dim == 512:
        assert not flagged.any() and sens.max() < 30
        assert dn[well].max() < 2e-5 and dw[well].max() < 5e-6
    else:
        assert 0.05 < flagged.mean() < 0.75
        assert dn[well].max() < 6e-5 and dw[well].max() < 5e-6, (dn[well].max(), dw[well].max())
    # the worst unflagged stamp against the oracle
    tabs = O.ao_tables(H, bool(three[k[0]]), 1, exact_masks=True)
    ofit, _ = O.compute_psf(lb[[k[1]]], see[k[0]], gl[k[0]], l0[k[0]], 1, H, bool(three[k[0]]), dim=dim, pixscale=ps, tables=tabs)
    record_margin('ill_conditioned_rule_dim%d' % dim, beta_vs_oracle=abs(a[k][4] - ofit[0, 4]))
    assert abs(a[k][4] - ofit[0, 4]) < 1e-4 and abs(b[k][4] - ofit[0, 4]) < 4e-5


def _oracle_rows(api, lb, see, gl, l0, three, dim, ps, npl, rows, lam_idx):
    """Oracle (fit, final stamps) for a sample of rows at a sample of wavelengths."""
    out = {}
    tabs = {t: O.ao_tables(H, bool(t), npl, exact_masks=True) for t in set(int(three[k]) for k in rows)}
    for k in rows:
        out[k] = O.compute_psf(lb[lam_idx], see[k], gl[k], l0[k], npl, H, bool(three[k]), dim=dim,
                               pixscale=ps, tables=tabs[int(three[k])])
    return out


def test_config_1000_rows_512(api):
    """BASELINE.json configs[2] on one GPU (its shard layout is exercised by tests/test_dist.py
    and tests/test_gpu_dist.py): 1000 rows x 35 wavelengths x 512^2 = 35 000 PSFs through one
    call -- two chunks of 500 on the two lanes, then nine chunks of at most 117 -- with the oracle on
    sampled rows and size-independent properties on everything."""
    n = 1000
    see, gl, l0 = api.synthetic_rows(n)
    lb = np.linspace(465, 930, 35)
    ps = api.grid_pixscale(512)
    three = (np.arange(n) % 97 == 5).astype(np.uint8)
    ctx = api.Context(dim=512, pixscale=ps, precision='mixed')
    r = ctx.reconstruct(lb, see, gl, l0, three, H)
    psf, fit = r['psf'], r['fit']
    assert np.isfinite(psf).all() and np.isfinite(fit).all()
    np.testing.assert_allclose(r['psf_sum'], psf.sum(axis=0), rtol=1e-12)
    assert np.all(fit[:, :, 14] == 0) and np.abs(fit[:, :, 1:3] - 20).max() < 0.1
    assert np.all(fit[:, -1, 5] < fit[:, 0, 5])
    # the first 100 rows are the bench workload: same bits as a 100-row call (chunking invariance)
    r100 = ctx.reconstruct(lb, see[:100], gl[:100], l0[:100], three[:100], H, want_psf=False)
    assert np.array_equal(r100['fit'], fit[:100])
    ctx.set_option('chunk_tasks', 117)          # the chunking of rounds 1-2: 8 x 117 + 64 rows
    r9 = ctx.reconstruct(lb, see, gl, l0, three, H, want_psf=False)
    assert np.array_equal(r9['fit'], fit)
    np.testing.assert_allclose(r9['psf_sum'], r['psf_sum'], rtol=1e-13)
    ctx.close()
    li = [0, 17, 34]
    for k, (ofit, ofin) in _oracle_rows(api, lb, see, gl, l0, three, 512, ps, 1, [5, 499, 999], li).items():
        assert rel_err(psf[k][li], ofin) < 2e-5, k
        assert np.abs(fit[k][li, 5] * ps - ofit[:, 3]).max() < 1e-4, k
        assert np.abs(fit[k][li, 4] - ofit[:, 4]).max() < 1e-4, k


def test_config_nine_directions_256(api):
    """BASELINE.json configs[3]: 100 rows x 35 wavelengths x 256^2, npsflin = 3 (nine evaluation
    directions per PSF, unweighted PSF mean -- SURVEY.md 8(d) item 3), through the SPARTA front
    end with the four LGS columns jittered by +-5 % and mean_of_lgs=True; the oracle on sampled
    rows (stamps: at 256^2 the fit is ill-posed)."""
    from collections import OrderedDict
    from muse_psfr_amd.psfrec import _table_hdu
    from muse_psfr_amd import _minifits as mf
    n = 100
    see, gl, l0 = api.synthetic_rows(n)
    rng = np.random.default_rng(77)
    cols = OrderedDict()
    for q in range(1, 5):
        j = 1 + 0.05 * rng.normal(size=(3, n))
        cols['LGS%d_SEEING' % q] = see * j[0]
        cols['LGS%d_TUR_GND' % q] = np.clip(gl * j[1], 0.05, 0.98)
        cols['LGS%d_L0' % q] = np.clip(l0 * j[2], 8.5, 29.5)
    tbl = _table_hdu(cols, {}, 'SPARTA_ATM_DATA')
    hdul = mf.HDUList([mf.PrimaryHDU(), tbl]) if isinstance(tbl, mf.BinTableHDU) else None
    if hdul is None:
        from astropy.io import fits
        hdul = fits.HDUList([fits.PrimaryHDU(), tbl])
    ps = api.grid_pixscale(256)
    lb = np.linspace(465, 930, 35)
    res = api.compute_psf_from_sparta(hdul, npsflin=3, lbda=lb, dim=256, pixscale=ps,
                                      cutoff_masks='exact', verbose=False)
    fr = res['FIT_ROWS'].data
    assert len(fr) == n * 35 and np.all(np.asarray(fr['lgs_idx']) == -1)
    vals = np.array([[cols['LGS%d_%s' % (q, c)] for c in ('SEEING', 'TUR_GND', 'L0')] for q in range(1, 5)])
    mean = vals.mean(axis=0)                               # (3, n): psfrec.py:1067
    np.testing.assert_allclose(np.asarray(fr['SEEING'])[::35], mean[0], rtol=1e-13)
    np.testing.assert_allclose(np.asarray(fr['L0'])[::35], mean[2], rtol=1e-13)
    # the same tasks through the C ABI with the stamps kept, against the oracle
    ctx = api.Context(dim=256, pixscale=ps, precision='mixed')
    r = ctx.reconstruct(lb, mean[0], mean[1], mean[2], np.zeros(n, np.uint8), H, npsflin=3)
    ctx.close()
    np.testing.assert_allclose(np.asarray(res['PSF_MEAN'].data), r['psf_sum'] / n, rtol=1e-12)
    np.testing.assert_array_equal(np.asarray(fr['n']), r['fit'][:, :, 4].reshape(-1))
    li = [0, 20, 34]
    for k, (_, ofin) in _oracle_rows(api, lb, mean[0], mean[1], mean[2], np.zeros(n, np.uint8), 256, ps, 3,
                                     [0, 63], li).items():
        assert rel_err(r['psf'][k][li], ofin) < 2e-5, k


def test_config_high_resolution_1024(api, golden, ref_masks):
    """BASELINE.json configs[4]: 200 rows x 70 wavelengths x 1024^2 = 14 000 PSFs.  Row 0 is the
    G6 golden row of the patched reference at 465 and 930 nm; the oracle on another sampled row;
    properties on everything."""
    n = 200
    see, gl, l0 = api.synthetic_rows(n)
    lb = np.linspace(465, 930, 70)
    ps = api.grid_pixscale(1024)
    ctx = api.Context(dim=1024, pixscale=ps, precision='mixed')
    r = ctx.reconstruct(lb, see, gl, l0, np.zeros(n, np.uint8), H, masks=ref_masks)
    ctx.close()
    psf, fit = r['psf'], r['fit']
    assert np.isfinite(psf).all() and np.isfinite(fit).all() and np.all(fit[:, :, 14] == 0)
    np.testing.assert_allclose(r['psf_sum'], psf.sum(axis=0), rtol=1e-12)
    assert np.all(fit[:, -1, 5] < fit[:, 0, 5])
    g = golden('g6_grids')
    gin = g['n1024_r1_in']
    assert (gin[0], gin[1], gin[2]) == (see[0], gl[0], l0[0]) and int(gin[3]) == 1 and int(gin[4]) == 0
    glb = g['n1024_lbda']
    for gi, li in ((0, 0), (3, 69)):
        assert glb[gi] == lb[li]
        assert rel_err(psf[0][li], g['n1024_r1_fin'][gi]) < 2e-5
        assert abs(fit[0][li, 5] * ps - g['n1024_r1_fit'][gi, 3]) < 1e-4
        assert abs(fit[0][li, 4] - g['n1024_r1_fit'][gi, 4]) < 1e-4
    tabs = O.ao_tables(H, False, 1, masks=ref_masks)
    li = [1, 35]
    ofit, ofin = O.compute_psf(lb[li], see[150], gl[150], l0[150], 1, H, False, dim=1024, pixscale=ps,
                               tables=tabs)
    assert rel_err(psf[150][li], ofin) < 2e-5
    assert np.abs(fit[150][li, 5] * ps - ofit[:, 3]).max() < 1e-4
    assert np.abs(fit[150][li, 4] - ofit[:, 4]).max() < 1e-4


@pytest.mark.parametrize('dim,npl', [(512, 1), (256, 3), (1024, 1), (128, 1)])
def test_line_pruning_changes_nothing_above_its_bound(api, dim, npl):
    """Mixed mode skips the lines of the OTF half plane whose elements are all below
    eps / (element count) of the PSF peak (option prune_eps, default 1e-9): against the same run
    with every line transformed no stamp pixel may move by more than eps of the peak (fp32
    rounding of the sums on top), the fits agree far inside the parity tolerance, and the number
    of lines kept grows with the wavelength."""
    see, gl, l0 = api.synthetic_rows(12)
    see[0], gl[0], l0[0] = 0.4, 0.95, 29.0            # sharpest PSF of the distribution: least pruning
    see[1], gl[1], l0[1] = 1.6, 0.30, 9.0             # broadest: most pruning
    lb = np.linspace(465, 930, 7)
    ps = api.grid_pixscale(dim)
    three = (np.arange(12) % 5 == 2).astype(np.uint8)
    out = {}
    for key, eps in (('all', 0.0), ('pruned', None), ('loose', 1e-6)):
        ctx = api.Context(dim=dim, pixscale=ps, precision='mixed')
        ctx.set_option('tier_eps', 0)          # the eps rule alone (the precision tiers have their own test)
        if eps is not None:
            ctx.set_option('prune_eps', eps)
        out[key] = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
        if key == 'pruned':
            if npl == 1:        # one direction: the block-masked kernel runs, which needs no line bounds
                with pytest.raises(api.MpsfrError):
                    ctx.debug_fetch('vkeep', (12, 4))
                ctx.set_option('mf_kernel', 1)
                ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
            vk = ctx.debug_fetch('vkeep', (12, 4))
        ctx.close()
    a, b, c = out['all'], out['pruned'], out['loose']
    peak = a['psf'].max(axis=(2, 3), keepdims=True)
    assert (np.abs(b['psf'] - a['psf']) / peak).max() < 3e-7          # eps + fp32 summation order
    assert (np.abs(c['psf'] - a['psf']) / peak).max() < 3e-6
    well = a['fit'][:, :, 4] < 10
    assert np.abs(b['fit'][:, :, 5] - a['fit'][:, :, 5])[well].max(initial=0) * ps < 2e-6
    assert np.abs(b['fit'][:, :, 4] - a['fit'][:, :, 4])[well].max(initial=0) < 2e-5
    nv = dim // 2 + 1
    assert vk.min() >= 1 and vk.max() <= nv
    assert np.all(np.diff(vk, axis=1) >= 0)              # grows with the wavelength pair
    assert vk[1].max() < vk[0].max()                     # the broad PSF needs fewer lines
    assert vk[1, 0] < 0.25 * nv                          # and most of its half plane is dropped


@pytest.mark.parametrize('prec', ['mixed', 'f64'])
@pytest.mark.parametrize('dim,npl', [(128, 1), (256, 3), (512, 1), (512, 2), (1024, 1), (1280, 1)])
def test_series_form_of_stage_a_against_the_full_size_transforms(api, dim, npl, prec):
    """Two independent implementations of stage A held against each other: the series + patch form
    (stage_a2.hip: polynomial in 1/L0^2 from per-context tables + pruned in-register transforms of the
    corrected zone; option stage_a = 2) and the full-size fp64 transforms of the PSD (stage_a.hip;
    stage_a = 0).  The structure function agrees to its storage rounding (fp32 in mixed mode: both are
    fp64 values rounded once) / to 1e-14 of its maximum (f64 mode); L0 from the edge of the expansion
    (7 m) to infinity, both laser geometries, fewer tasks than waves; a call with L0 < 7 m takes the
    full-size transforms by itself, whatever the option says."""
    see = np.array([0.45, 1.0, 1.6, 0.8, 1.2])
    gl = np.array([0.93, 0.7, 0.3, 0.5, 0.05])
    l0 = np.array([7.0, 25.0, 1.0e4, 29.9, 8.1])
    three = np.array([0, 1, 0, 0, 1], np.uint8)
    lb = np.array([465.0, 700.0, 930.0]) if dim != 1280 else np.array([490.0, 700.0, 930.0])
    ps = api.grid_pixscale(dim)
    ndir = npl * npl
    res = {}
    for mode in (0, 2):
        ctx = api.Context(dim=dim, pixscale=ps, precision=prec)
        ctx.set_option('stage_a', mode)
        r = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
        d0 = ctx.debug_fetch('dphi0', (see.size, ndir, dim // 2 + 1, dim))
        tel = ctx.debug_fetch('tel', (dim // 2 + 1, dim))
        if mode == 2:       # an outer scale below the radius of the expansion: the call falls back by itself
            r_short = ctx.reconstruct(lb, see[:2], gl[:2], np.array([5.0, 25.0]), three[:2], H, npsflin=npl)
        ctx.close()
        res[mode] = (r, d0)
    ctx = api.Context(dim=dim, pixscale=ps, precision=prec)
    ctx.set_option('stage_a', 0)
    r_short0 = ctx.reconstruct(lb, see[:2], gl[:2], np.array([5.0, 25.0]), three[:2], H, npsflin=npl)
    ctx.close()
    np.testing.assert_array_equal(r_short['psf'], r_short0['psf'])
    (ra, da), (rb, db) = res[0], res[2]
    tol_d = 1.5e-7 if prec == 'mixed' else 1e-14
    # The series form skips the pieces of a line (series_lanes(N) = 16 / 32 / 64 columns) on which the telescope OTF
    # vanishes identically: compared is the kept region; the skipped one holds the zeros the buffer was allocated
    # with, lies wholly outside the support (so the per-wavelength stage multiplies it by zero, psfrec.py:784-797),
    # and is most of what lies outside it (21 % of the half plane).
    inside = tel > 0
    L = 16 if dim <= 256 else (32 if dim == 512 else 64)
    zero = np.all(db == 0, axis=(0, 1))
    skipped = np.repeat(zero.reshape(dim // 2 + 1, dim // L, L).all(axis=2), L, axis=1)      # whole pieces only
    if dim >= 256:
        assert not np.any(skipped & inside)
        assert skipped.sum() > 0.6 * (~inside).sum(), (skipped.sum(), (~inside).sum())
    kept = np.broadcast_to(~skipped, da[0].shape)
    for k in range(see.size):
        err = np.abs(db[k] - da[k])[kept].max() / np.abs(da[k]).max()
        assert err < tol_d, (k, err)
    peak = ra['psf'].max(axis=(2, 3), keepdims=True)
    dst = float((np.abs(rb['psf'] - ra['psf']) / peak).max())
    assert dst < (5e-7 if prec == 'mixed' else 1e-12), dst
    if dim >= 512:
        well = ra['fit'][:, :, 4] < 10
        assert np.abs(rb['fit'][:, :, 5] - ra['fit'][:, :, 5])[well].max(initial=0) * ps < (5e-6 if prec == 'mixed' else 1e-8)
        assert np.abs(rb['fit'][:, :, 4] - ra['fit'][:, :, 4])[well].max(initial=0) < (2e-5 if prec == 'mixed' else 1e-8)
        record_margin('series_vs_full_size_%s' % prec, stamp=dst,
                      beta=float(np.abs(rb['fit'][:, :, 4] - ra['fit'][:, :, 4])[well].max(initial=0)))


TIER_EPS, PRUNE_EPS = 4.0e-6, 1.0e-9        # the library's defaults (include/mpsfr.h)


def _tier_runs(api, dim, ps, lb, see, gl, l0, variants):
    out = {}
    for key, opts in variants:
        ctx = api.Context(dim=dim, pixscale=ps, precision='mixed')
        for k, v in opts.items():
            ctx.set_option(k, v)
        out[key] = ctx.reconstruct(lb, see, gl, l0, None, H)
        out[key + '_pre'] = ctx.debug_fetch('pre', (len(see), len(lb), 40, 40))
        out[key + '_work'] = ctx.debug_fetch('mf_work', (5,))
        ctx.close()
    return out


def _tier_bound(pre, eps):
    """What include/mpsfr.h promises for a stamp before the convolutions: every pixel within eps of the stamp's
    peak, plus the stamp's normalisation to unit sum (psfrec.py:685) in the worst case that all 1600 pixels moved
    the same way: eps (1 + 1600 peak / sum), per stamp."""
    peak = pre.max(axis=(2, 3))
    return eps * (1 + 1600 * peak / pre.sum(axis=(2, 3)))


@pytest.mark.parametrize('case', ['bench512', 'sharp1280', 'plateau512'])
def test_precision_tiers_of_the_matrix_core_stage(api, case):
    """The matrix-core stage drops blocks below 2^-29 of OTF[0][0] ("mf_floor") and runs blocks below 2^-18
    without the low half of the OTF ("mf_mid_log2"), where the fp16 halves left out are subnormal (otf_mfma2.hip,
    DESIGN.md 2.9).  These are approximations (gfx950's matrix cores keep subnormal fp16 inputs), held to a budget
    per (task, wavelength): the OTF mass a tier leaves out is below tier_eps / 2 of a lower bound of the PSF peak,
    and K_MF_PREP lowers the thresholds where it would not be ("tier_eps", default 4e-6).  Asserted here: (1) the
    documented bound, per stamp, of the default path against the path without tiers -- on the bench rows, on the
    sharpest PSF the SPARTA filter admits at the native 1280^2 grid (seeing 0.3", GL 0.98, 930 nm: VERDICT r4 #3),
    and on rows whose OTF has its coherent plateau just below the floor; (2) the size the approximations really
    have on these workloads (3e-7 of the peak, fits far inside the parity tolerance); (3) that the budget acts:
    a small tier_eps executes more blocks and lands closer to the path without tiers."""
    if case == 'bench512':
        dim, ps = 512, api.grid_pixscale(512)
        see, gl, l0 = api.synthetic_rows(10)
        see[0], gl[0], l0[0] = 0.4, 0.95, 29.0
        see[1], gl[1], l0[1] = 1.6, 0.30, 9.0
        lb = np.linspace(465, 930, 9)
    elif case == 'sharp1280':
        dim, ps = 1280, 0.2
        see = np.array([0.3, 0.3, 0.35, 0.5])
        gl = np.array([0.98, 0.98, 0.9, 0.98])
        l0 = np.array([29.9, 8.1, 20.0, 15.0])
        lb = np.array([490.0, 600.0, 700.0, 800.0, 930.0])
    else:
        dim, ps = 512, api.grid_pixscale(512)
        see = np.array([0.6, 0.5, 0.3, 0.9, 0.4])
        gl = np.array([0.5, 0.9, 0.5, 0.9, 0.95])
        l0 = np.array([15.0, 20.0, 29.0, 11.0, 29.0])
        lb = np.linspace(465, 930, 7)
    out = _tier_runs(api, dim, ps, lb, see, gl, l0,
                     (('default', {}), ('no_tiers', {'tier_eps': 0}), ('no_floor', {'mf_floor': 0}),
                      ('no_mid', {'mf_mid_log2': -1e30}), ('tight', {'tier_eps': 1e-8}),
                      ('no_budget', {'tier_eps': float('inf')})))
    a, ref = out['default'], out['no_tiers']
    # (3) the switches do what they say
    assert out['no_tiers_work'][4] == 0 and out['no_mid_work'][4] == 0
    assert out['no_tiers_work'][0] >= out['tight_work'][0] >= out['default_work'][0] >= out['no_budget_work'][0]
    if case != 'sharp1280':
        assert out['default_work'][4] > 0 and out['no_floor_work'][0] > out['default_work'][0]
    # (1) the documented bound of the default path, per stamp, before the convolutions ...
    peak = out['no_tiers_pre'].max(axis=(2, 3))
    err = np.abs(out['default_pre'] - out['no_tiers_pre']).max(axis=(2, 3)) / peak
    bound = _tier_bound(out['no_tiers_pre'], TIER_EPS)
    assert np.all(err <= bound + 2e-7), (case, float((err / bound).max()))      # (+ the fp32 rounding of two runs)
    errt = np.abs(out['tight_pre'] - out['no_tiers_pre']).max(axis=(2, 3)) / peak
    assert np.all(errt <= _tier_bound(out['no_tiers_pre'], 1e-8) + 2e-7), (case, float(errt.max()))
    # ... and after them (non-negative kernels of unit sum: contractions in the maximum norm)
    fpeak = ref['psf'].max(axis=(2, 3))
    ferr = np.abs(a['psf'] - ref['psf']).max(axis=(2, 3)) / fpeak
    assert np.all(ferr <= bound * peak / fpeak * 1.0 + 4e-7), (case, float(ferr.max()))
    # (2) their real size
    well = ref['fit'][:, :, 4] < 10
    dst = float(ferr.max())
    dfw = float(np.abs(a['fit'][:, :, 5] - ref['fit'][:, :, 5])[well].max(initial=0) * ps)
    dbe = float(np.abs(a['fit'][:, :, 4] - ref['fit'][:, :, 4])[well].max(initial=0))
    record_margin('precision_tiers_' + case, stamp=dst, stamp_pre=float(err.max()), fwhm_arcsec=dfw, beta=dbe,
                  bound_used_fraction=float((err / bound).max()),
                  blocks_default=out['default_work'][0], blocks_no_tiers=out['no_tiers_work'][0],
                  blocks_no_budget=out['no_budget_work'][0])
    assert dst < 3e-7 and dfw < 2e-6 and dbe < 2e-5, (case, dst, dfw, dbe)


@pytest.mark.parametrize('dim,npl', [(512, 1), (128, 1), (256, 2), (256, 5), (512, 3), (1280, 1)])
def test_matrix_core_stage_against_the_fft_stage(api, dim, npl):
    """Mixed mode has two implementations of the per-wavelength stage: split-fp16 contractions on
    the matrix cores (default) and LDS FFTs (otf_mfma = 0).  Both are pinned to the oracle elsewhere; here they must agree with each other
    at fp32 level on stamps and fits, with and without pruning, and the work the matrix-core stage
    reports must shrink with the pruning."""
    see, gl, l0 = api.synthetic_rows(6)
    see[0], gl[0], l0[0] = 0.45, 0.9, 28.0
    lb = np.array([470., 600., 760., 925.]) if dim != 1280 else np.array([500., 700., 930.])
    ps = api.grid_pixscale(dim) if dim != 1280 else 0.2
    three = (np.arange(6) % 3 == 1).astype(np.uint8)
    out, work = {}, {}
    for key, opts in (('fft', {'otf_mfma': 0}), ('mfma', {}), ('mfma_all', {'prune_eps': 0.0})):
        ctx = api.Context(dim=dim, pixscale=ps, precision='mixed')
        for k, v in opts.items():
            ctx.set_option(k, v)
        out[key] = ctx.reconstruct(lb, see, gl, l0, three, H, npsflin=npl)
        if key != 'fft':
            work[key] = ctx.debug_fetch('mf_work', (3,))
        else:
            with pytest.raises(api.MpsfrError):
                ctx.debug_fetch('mf_work', (3,))
        ctx.close()
    a = out['fft']
    peak = a['psf'].max(axis=(2, 3), keepdims=True)
    for key in ('mfma', 'mfma_all'):
        b = out[key]
        assert (np.abs(b['psf'] - a['psf']) / peak).max() < 5e-6, key
        well = a['fit'][:, :, 4] < 10
        assert np.abs(b['fit'][:, :, 5] - a['fit'][:, :, 5])[well].max(initial=0) * ps < 1e-5, key
        assert np.abs(b['fit'][:, :, 4] - a['fit'][:, :, 4])[well].max(initial=0) < 5e-5, key
    assert work['mfma_all'][0] == work['mfma_all'][2]            # no pruning: every tile step
    assert 0 < work['mfma'][0] < 0.8 * work['mfma_all'][0]
    assert work['mfma'][1] <= work['mfma_all'][1]


def test_line_pruning_in_f64_mode(api):
    """f64 mode prunes the trailing lines of the half plane as well, with its own bound
    (prune_eps_f64, default 1e-13 of the PSF peak -- two decades under the 1e-11 its stamps reach
    against the oracle): no stamp pixel moves by more than that, most of a broad PSF's half plane
    is dropped, and the mixed-mode option keeps its own range."""
    see, gl, l0 = np.array([0.45, 1.6, 1.0]), np.array([0.9, 0.3, 0.7]), np.array([28.0, 9.0, 25.0])
    lb = np.array([465.0, 700.0, 930.0])
    ps = api.grid_pixscale(512)
    out = {}
    for key, eps in (('pruned', None), ('all', 0.0)):
        ctx = api.Context(dim=512, pixscale=ps, precision='f64')
        if eps is not None:
            ctx.set_option('prune_eps_f64', eps)
        out[key] = ctx.reconstruct(lb, see, gl, l0, np.zeros(3, np.uint8), H)
        if key == 'pruned':
            vk = ctx.debug_fetch('vkeep', (3, 2))
            with pytest.raises(api.MpsfrError):
                ctx.set_option('prune_eps', 0.1)
            with pytest.raises(api.MpsfrError):
                ctx.set_option('prune_eps_f64', 1e-3)
        else:
            with pytest.raises(api.MpsfrError):
                ctx.debug_fetch('vkeep', (3, 2))
        ctx.close()
    a, b = out['all'], out['pruned']
    peak = a['psf'].max(axis=(2, 3), keepdims=True)
    assert (np.abs(b['psf'] - a['psf']) / peak).max() < 2e-13
    assert np.abs(b['fit'][:, :, 4] - a['fit'][:, :, 4]).max() < 1e-9
    assert vk.min() >= 1 and vk.max() <= 257 and np.all(np.diff(vk, axis=1) >= 0)
    assert vk[1, 0] < 0.35 * 257 and vk[1].max() < vk[0].max()


@pytest.mark.parametrize('opts', [{}, {'mf_kernel': 1}, {'otf_mfma': 0}])
def test_wavelengths_in_any_order(api, opts):
    """The reference takes the wavelengths in any order (psfrec.py:667 loops over them as given); the
    pruning of the per-wavelength stage bounds pairs and groups of wavelengths by their longest
    member, which must not be assumed to be the last one."""
    see, gl, l0 = api.synthetic_rows(5)
    see[0], gl[0], l0[0] = 0.45, 0.9, 28.0
    lb = np.linspace(465, 930, 11)
    perm = np.array([7, 0, 10, 3, 5, 1, 9, 2, 8, 4, 6])
    ps = api.grid_pixscale(512)
    res = []
    for order in (np.arange(11), perm, np.arange(11)[::-1]):
        ctx = api.Context(dim=512, pixscale=ps, precision='mixed')
        for k, v in opts.items():
            ctx.set_option(k, v)
        r = ctx.reconstruct(lb[order], see, gl, l0, np.zeros(5, np.uint8), H)
        ctx.close()
        inv = np.argsort(order)
        res.append((r['psf'][:, inv], r['fit'][:, inv]))
    peak = res[0][0].max(axis=(2, 3), keepdims=True)
    for psf, fit in res[1:]:
        if 'otf_mfma' in opts:      # two wavelengths share a complex transform: the pairing changes the rounding
            assert (np.abs(psf - res[0][0]) / peak).max() < 2e-6
            assert np.abs(fit[:, :, 4] - res[0][1][:, :, 4]).max() < 2e-5
        else:                       # every stamp is computed on its own: bit for bit
            np.testing.assert_array_equal(psf, res[0][0])
            np.testing.assert_array_equal(fit, res[0][1])


def test_automatic_chunking_keeps_a_call_in_one_pass(api):
    """An asynchronous call (device outputs; its neighbours run on the other lane) is one pipeline pass
    up to 512 tasks and balanced passes (a multiple of the two lanes) beyond: 125 rows -- the 8-GPU
    shard of BASELINE.json configs[2] -- must not run as 118 + 7.  A synchronous call (host outputs)
    of 8192 stamps or more runs as one pass per lane.  (mpsfr_debug_fetch hands out the LAST pass, whose
    size is what is checked here.)"""
    import torch
    lb = np.linspace(500.0, 900.0, 16)          # (with few wavelengths a pass may be larger: >= 4096 stamps)
    ps = api.grid_pixscale(128)
    ctx = api.Context(dim=128, pixscale=ps, precision='mixed')
    dev = torch.device('cuda', 0)
    for n, last_async, last_sync in ((125, 125, 125), (511, 511, 511), (512, 512, 256), (513, 256, 256),
                                     (1100, 275, 275)):      # 512 rows x 16 = 8192 stamps; 513 = 257 + 256
        see, gl, l0 = api.synthetic_rows(n)
        r = ctx.reconstruct(lb, see, gl, l0, np.zeros(n, np.uint8), H, want_psf=False)
        assert np.isfinite(r['fit']).all()
        pre = ctx.debug_fetch('pre', (last_sync, 16, 40, 40))   # raises unless the last pass had that many tasks
        assert pre.shape[0] == last_sync
        fit = torch.zeros((n, 16, api.NFIT), dtype=torch.float64, device=dev)
        psum = torch.zeros((16, 40, 40), dtype=torch.float64, device=dev)
        ctx.reconstruct_device(lb, see, gl, l0, np.zeros(n, np.uint8), H, 12.0, 1, None, None,
                               psum.data_ptr(), fit.data_ptr())
        ctx.sync()
        pre = ctx.debug_fetch('pre', (last_async, 16, 40, 40))
        assert pre.shape[0] == last_async
        np.testing.assert_array_equal(fit.cpu().numpy(), r['fit'])          # the passes do not change a fit
        np.testing.assert_allclose(psum.cpu().numpy(), r['psf_sum'], rtol=1e-13)
    ctx.close()
