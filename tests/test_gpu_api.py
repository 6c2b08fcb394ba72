"""GPU: the drop-in Python API (muse_psfr_amd.compute_psf / compute_psf_from_sparta) replaying the
reference's own integration tests (muse_psfr/test_psfrec.py) with their known answers."""
import logging
import os

import numpy as np
import pytest

from conftest import H, rel_err, record_margin
import psfr_oracle as O

pytestmark = pytest.mark.gpu


_REF_MASKS = []


@pytest.fixture(scope='module')
def api(ref_masks):
    import muse_psfr_amd
    _REF_MASKS.append(ref_masks)      # the CLI tests patch the host masks with the golden ones
    return muse_psfr_amd


def _hdul(tbl):
    from muse_psfr_amd import _minifits as mf
    from muse_psfr_amd.psfrec import _astropy
    fits, _ = _astropy()
    if fits is not None:
        return fits.HDUList([fits.PrimaryHDU(), tbl])
    return mf.HDUList([mf.PrimaryHDU(), tbl])


def test_reconstruction(api, ref_masks):
    """test_psfrec.py:17-30 (npsflin=3, 5 wavelengths)."""
    res = api.compute_psf_from_sparta(_hdul(api.create_sparta_table()), npsflin=3, lmin=490,
                                      lmax=541.76, nl=5, cutoff_masks=ref_masks)
    assert len(res) == 5
    fit = res['FIT_ROWS'].data
    np.testing.assert_allclose(fit['L0'], 25)
    np.testing.assert_allclose(fit['center'], 20, atol=1e-3)
    np.testing.assert_allclose(fit[1]['lbda'], 502.9, atol=1e-1)
    np.testing.assert_allclose(fit[1]['fwhm'], 0.85, atol=1e-2)


def test_fit_poly(api, ref_masks):
    """test_psfrec.py:33-44."""
    res = api.compute_psf_from_sparta(_hdul(api.create_sparta_table()), lmin=500, lmax=900, nl=9,
                                      cutoff_masks=ref_masks)
    fit = res['FIT_ROWS'].data
    r = api.fit_psf_with_polynom(fit['lbda'], fit['fwhm'][:, 0], fit['n'], deg=(5, 5), output=1)
    np.testing.assert_allclose(r['fwhm_pol'][0], 0.65, atol=1e-2)
    np.testing.assert_allclose(r['beta_pol'][0], 0.78, atol=1e-2)
    np.testing.assert_allclose(r['beta_fit'][8], fit[1]['n'], atol=1e-2)
    np.testing.assert_allclose(r['fwhm_fit'][8], fit[1]['fwhm'][0], atol=1e-2)


def test_reconstruction2(api, ref_masks):
    """test_psfrec.py:47-69 (mean_of_lgs=False, one rejected laser, npsflin=3)."""
    tbl = api.create_sparta_table()
    tbl.data[0]['LGS1_L0'] = 20
    tbl.data[0]['LGS1_SEEING'] = 0.8
    tbl.data[0]['LGS1_TUR_GND'] = 0.5
    tbl.data[0]['LGS3_L0'] = 100
    res = api.compute_psf_from_sparta(_hdul(tbl), npsflin=3, lmin=500, lmax=700, nl=3,
                                      mean_of_lgs=False, cutoff_masks=ref_masks)
    assert len(res) == 5
    fit = res['FIT_ROWS'].data
    np.testing.assert_allclose(fit[fit['lgs_idx'] == 1]['L0'], 20)
    np.testing.assert_allclose(fit[fit['lgs_idx'] != 1]['L0'], 25)
    np.testing.assert_allclose(fit['center'], 20, atol=1e-3)
    np.testing.assert_allclose(fit[fit['lbda'] == 500]['fwhm'][:, 0], [0.79, 0.86, 0.86], atol=1e-2)


def test_bad_l0(api, ref_masks, tmp_path, caplog):
    """test_psfrec.py:72-90 (file input, LGS4_L0 = 150 -> three-laser mode, log strings)."""
    testfile = os.path.join(str(tmp_path), 'sparta.fits')
    api.create_sparta_table(outfile=testfile, bad_l0=True)
    with caplog.at_level(logging.INFO, logger='muse_psfr_amd.psfrec'):
        res = api.compute_psf_from_sparta(testfile, lmin=490, lmax=541.76, nl=5,
                                          cutoff_masks=ref_masks)
    assert caplog.records[1].message == '1/1 : Using only 3 values out of 4 after outliers rejection'
    assert caplog.records[3].message == 'Using three lasers mode'
    assert len(res) == 5
    fit = res['FIT_ROWS'].data
    np.testing.assert_allclose(fit['L0'], 25)
    np.testing.assert_allclose(fit['center'], 20, atol=1e-3)
    np.testing.assert_allclose(fit[1]['lbda'], 502.9, atol=1e-1)
    np.testing.assert_allclose(fit[1]['fwhm'], 0.86, atol=1e-2)
    out = os.path.join(str(tmp_path), 'out.fits')
    res.writeto(out, overwrite=True)
    assert os.path.getsize(out) > 0


def test_cli_table_values(api, ref_masks):
    """test_psfrec.py:121-127: --values 1,0.7,25 -> LBDA 5000 7000 9000 / FWHM 0.85 0.73 0.62 /
    BETA 2.73 2.55 2.23, read from FIT_MEAN exactly as cli.py:66-69 does."""
    res = api.compute_psf_from_sparta(_hdul(api.create_sparta_table(seeing=1, GL=0.7, L0=25)),
                                      lmin=500, lmax=900, nl=3, cutoff_masks=ref_masks)
    assert [h.name for h in res] == ['PRIMARY', 'SPARTA_ATM_DATA', 'FIT_ROWS', 'FIT_MEAN', 'PSF_MEAN']
    data = res['FIT_MEAN'].data
    hdr = res['FIT_MEAN'].header
    assert (hdr['SEEING'], hdr['GL'], hdr['L0']) == (1.0, 0.7, 25.0)
    assert 'LBDA %.0f %.0f %.0f' % tuple(data['lbda'] * 10) == 'LBDA 5000 7000 9000'
    assert 'FWHM %.2f %.2f %.2f' % tuple(data['fwhm'][:, 0]) == 'FWHM 0.85 0.73 0.62'
    assert 'BETA %.2f %.2f %.2f' % tuple(data['n']) == 'BETA 2.73 2.55 2.23'


def test_compute_psf_against_golden(api, golden, ref_masks):
    g = golden('g2_native1280')
    tbl, psf = api.compute_psf(g['lbda'], 1.0, 0.7, 25.0, cutoff_masks=ref_masks, verbose=False)
    assert psf.shape == (5, 40, 40)
    assert np.abs(psf - g['fin_0']).max() / g['fin_0'].max() < 2e-5
    assert np.abs(np.asarray(tbl['fwhm'])[:, 0] - g['fit_0'][:, 3]).max() < 1e-4
    assert np.abs(np.asarray(tbl['n']) - g['fit_0'][:, 4]).max() < 1e-4
    assert np.all(np.asarray(tbl['SEEING']) == 1.0) and tbl.meta['L0'] == 25.0
    # the default ('host') masks are those of this machine's NumPy: self-consistent with the oracle
    import psfr_oracle as O
    tbl2, psf2 = api.compute_psf([500.0, 900.0], 1.2, 0.6, 18.0, verbose=False)
    tabs = O.ao_tables(H, False, 1, masks=O.numpy_cutoff_masks())
    ofit, ofin = O.compute_psf([500.0, 900.0], 1.2, 0.6, 18.0, tables=tabs)
    assert np.abs(psf2 - ofin).max() / ofin.max() < 2e-5
    assert np.abs(np.asarray(tbl2['n']) - ofit[:, 4]).max() < 1e-4
    with pytest.raises(ValueError):                      # psfrec.py:663-683 at the native grid
        api.compute_psf([465.0], 1.0, 0.7, 25.0, verbose=False)


def test_script(api, tmp_path, caplog, monkeypatch):
    """test_psfrec.py:103-144: `--values 1,0.7,25` -> the three-wavelength table in the log file
    and in the log records; `--values 1,0.7,1000` -> 'No results'."""
    from muse_psfr_amd import cli, psfrec
    monkeypatch.setattr(psfrec, 'host_cutoff_masks', lambda: _REF_MASKS[0])
    with pytest.raises(SystemExit, match='No results'):
        cli.main(['--values', '1,0.7,1000'])
    caplog.clear()
    logfile = os.path.join(str(tmp_path), 'muse-psfr2.log')
    with caplog.at_level(logging.INFO, logger='muse_psfr_amd'):
        cli.main(['--no-color', '--values', '1,0.7,25', '--logfile', logfile])
    lines = open(logfile).read().splitlines()
    assert lines[2:] == ['-' * 68, 'Sparta Seeing: 1.00 arcsec GL: 0.70 L0:25.00 m',
                         'LBDA 5000 7000 9000', 'FWHM 0.85 0.73 0.62', 'BETA 2.73 2.55 2.23', '-' * 68]
    records = [r for r in caplog.records if r.levelname != 'DEBUG']
    assert records[6].message == 'LBDA 5000 7000 9000'
    assert records[7].message == 'FWHM 0.85 0.73 0.62'
    assert records[8].message == 'BETA 2.73 2.55 2.23'


def test_script_with_file(api, tmp_path, monkeypatch):
    """test_psfrec.py:147-170: file input, log-file text, output FITS extensions."""
    from muse_psfr_amd import cli, psfrec, _minifits
    monkeypatch.setattr(psfrec, 'host_cutoff_masks', lambda: _REF_MASKS[0])
    testfile = os.path.join(str(tmp_path), 'sparta.fits')
    api.create_sparta_table(outfile=testfile)
    logfile = os.path.join(str(tmp_path), 'muse_psfr.log')
    outfile = os.path.join(str(tmp_path), 'out.fits')
    cli.main([testfile, '--no-color', '--logfile', logfile, '--outfile', outfile])
    fits, _ = psfrec._astropy()
    hdul = (fits or _minifits).open(outfile)
    assert [h.name for h in hdul] == ['PRIMARY', 'SPARTA_ATM_DATA', 'FIT_ROWS', 'FIT_MEAN', 'PSF_MEAN']
    lines = open(logfile).read().splitlines()
    assert lines[2:] == ['OB None None Airmass 0.00-0.00', '-' * 68,
                         'Sparta Seeing: 1.00 arcsec GL: 0.70 L0:25.00 m', 'LBDA 5000 7000 9000',
                         'FWHM 0.85 0.73 0.62', 'BETA 2.73 2.55 2.23', '-' * 68]


def test_plot(api, tmp_path):
    """test_psfrec.py:173-188: 2 rows, the default 35 wavelengths on the native grid, plot_psf on
    the result and on the file written from it (Agg backend)."""
    import matplotlib
    matplotlib.use('agg', force=True)
    testfile = os.path.join(str(tmp_path), 'sparta.fits')
    api.create_sparta_table(outfile=testfile, nlines=2)
    res = api.compute_psf_from_sparta(testfile)
    assert res['PSF_MEAN'].data.shape == (35, 40, 40)
    assert len(res['FIT_ROWS'].data) == 70
    outfile = os.path.join(str(tmp_path), 'fitres.fits')
    res.writeto(outfile, overwrite=True)
    fig = api.plot_psf(res)
    fig.savefig(os.path.join(str(tmp_path), 'fig.png'))
    fig = api.plot_psf(outfile)
    fig.savefig(os.path.join(str(tmp_path), 'fig.png'))
    assert os.path.getsize(os.path.join(str(tmp_path), 'fig.png')) > 0
    # `plot=True` path of compute_psf_from_sparta (psfrec.py:1115-1118) on a non-interactive backend
    res2 = api.compute_psf_from_sparta(testfile, lmin=500, lmax=900, nl=3, plot=True)
    assert len(res2) == 5


@pytest.mark.parametrize('tag,mean', [('mean', True), ('lgs', False)])
def test_sparta_front_end_against_the_reference(api, golden, ref_masks, tag, mean):
    """G7: the reference's own compute_psf_from_sparta end to end (psfrec.py:981-1120) on a table
    with the four LGS columns jittered by +-5 % (BASELINE.json configs[3]), one rejected laser
    (3-LGS mode) and one row without a valid laser: FIT_ROWS incl. row_idx / lgs_idx, FIT_MEAN,
    PSF_MEAN, for mean_of_lgs=True and mean_of_lgs=False (psfrec.py:1066-1076).

    The `flux` and `err_*` columns of G7 are DEFINED BY THE ORACLE, parity unpinned: mpdaf is not
    importable, so oracle/_refload.py plugs the oracle's moffat_fit(errors=True) -- the restatement of mpdaf's
    covariance recipe -- into the reference's fit_psf_cube; the 2e-3 checks on these columns hold the library
    against that restatement, not against reference output.  fwhm / n / center / peak: MINPACK to 1e-7 and the
    reference's known answers to 1e-2."""
    from collections import OrderedDict
    from muse_psfr_amd.psfrec import _table_hdu
    g = golden('g7_sparta_lgs')
    cols = OrderedDict((str(n), np.array(v)) for n, v in zip(g['colnames'], g['table']))
    tbl = _table_hdu(cols, {}, 'SPARTA_ATM_DATA')
    res = api.compute_psf_from_sparta(_hdul(tbl), lmin=float(g['lmin']), lmax=float(g['lmax']),
                                      nl=int(g['nl']), mean_of_lgs=mean, cutoff_masks=ref_masks,
                                      verbose=False)
    fr, fm = res['FIT_ROWS'].data, res['FIT_MEAN'].data
    assert np.array_equal(np.asarray(fr['row_idx']), g[tag + '_rows_row_idx'])
    assert np.array_equal(np.asarray(fr['lgs_idx']), g[tag + '_rows_lgs_idx'])
    for c in ('lbda', 'SEEING', 'GL', 'L0'):
        np.testing.assert_allclose(np.asarray(fr[c]), g[tag + '_rows_' + c], rtol=1e-13)
    assert np.abs(np.asarray(fr['fwhm']) - g[tag + '_rows_fwhm']).max() < 1e-4
    assert np.abs(np.asarray(fr['n']) - g[tag + '_rows_n']).max() < 1e-4
    assert np.abs(np.asarray(fr['center']) - g[tag + '_rows_center']).max() < 1e-4
    assert np.abs(np.asarray(fr['peak']) / g[tag + '_rows_peak'] - 1).max() < 1e-4
    assert np.abs(np.asarray(fm['fwhm']) - g[tag + '_mean_fwhm']).max() < 1e-4
    assert np.abs(np.asarray(fm['n']) - g[tag + '_mean_n']).max() < 1e-4
    # the columns psfrec.py:866-870 keeps besides (fwhm, n, center, peak): flux and the err_* of mpdaf's
    # recipe (G7 holds them as the reference's own table carries them with the oracle's fit plugged in)
    np.testing.assert_allclose(np.asarray(fr['flux']), g[tag + '_rows_flux'], rtol=1e-4)
    np.testing.assert_allclose(np.asarray(fm['flux']), g[tag + '_mean_flux'], rtol=1e-4)
    for c in ('err_center', 'err_flux', 'err_fwhm', 'err_n', 'err_peak'):
        np.testing.assert_allclose(np.asarray(fr[c]), g['%s_rows_%s' % (tag, c)], rtol=2e-3, err_msg=c)
        np.testing.assert_allclose(np.asarray(fm[c]), g['%s_mean_%s' % (tag, c)], rtol=2e-3, err_msg=c)
    hdr = res['FIT_MEAN'].header
    np.testing.assert_allclose([hdr['SEEING'], hdr['GL'], hdr['L0']], g[tag + '_mean_hdr'], rtol=1e-13)
    pm = np.asarray(res['PSF_MEAN'].data)
    assert np.abs(pm - g[tag + '_psf_mean']).max() / g[tag + '_psf_mean'].max() < 2e-5


def test_fan_out_over_devices_matches_one_context(api, ref_masks):
    """compute_psf_from_sparta(devices=[...]): row shards on one context and one host thread per
    device -- the reference's n_jobs fan-out (psfrec.py:1082-1083, 1104-1105).  Two contexts on
    the one GPU of the test box: FIT_ROWS bit for bit, PSF_MEAN to the summation order."""
    from muse_psfr_amd import psfrec
    rng = np.random.default_rng(11)
    tbl = api.create_sparta_table(nlines=37)
    for k in range(1, 5):                       # 37 different rows, one with a rejected laser
        tbl.data['LGS%d_SEEING' % k][:] = rng.uniform(0.5, 1.4, 37)
        tbl.data['LGS%d_TUR_GND' % k][:] = rng.uniform(0.2, 0.9, 37)
        tbl.data['LGS%d_L0' % k][:] = rng.uniform(10.0, 28.0, 37)
    tbl.data['LGS4_L0'][5] = 150.0
    kw = dict(lmin=500, lmax=900, nl=4, cutoff_masks=ref_masks, verbose=False, dim=256,
              pixscale=api.grid_pixscale(256))
    one = api.compute_psf_from_sparta(_hdul(tbl), devices=[0], **kw)
    two = api.compute_psf_from_sparta(_hdul(tbl), devices=[0, 0], **kw)
    three = api.compute_psf_from_sparta(_hdul(tbl), devices=[0, 0, 0], mean_of_lgs=False, **kw)
    for c in ('lbda', 'fwhm', 'n', 'center', 'peak', 'flux', 'row_idx', 'lgs_idx'):
        np.testing.assert_array_equal(np.asarray(one['FIT_ROWS'].data[c]), np.asarray(two['FIT_ROWS'].data[c]))
    np.testing.assert_allclose(np.asarray(two['PSF_MEAN'].data), np.asarray(one['PSF_MEAN'].data), rtol=1e-13)
    np.testing.assert_allclose(np.asarray(two['FIT_MEAN'].data['n']), np.asarray(one['FIT_MEAN'].data['n']), rtol=1e-9)
    assert len(three['FIT_ROWS'].data) == (37 * 4 - 1) * 4
    # the automatic choice: one visible GPU, or n_jobs = 1, keeps one device
    assert psfrec._fanout_devices(None, None, 1000, 1) == [0]
    assert psfrec._fanout_devices(None, None, 10, -1) == [0]
    assert psfrec._fanout_devices([1, 1], 0, 3, -1) == [1, 1]


def test_reconstruct_multi_in_the_library(api):
    """mpsfr_reconstruct_multi: row shards over several contexts with one host thread each inside the
    library (here three contexts on the one GPU) against the single-context call; an error in a
    worker thread reaches the caller; fewer rows than contexts."""
    from muse_psfr_amd._lib import Context, MpsfrError
    see, gl, l0 = api.synthetic_rows(23)
    three = (np.arange(23) % 4 == 0).astype(np.uint8)
    lb = np.linspace(500, 900, 5)
    ps = api.grid_pixscale(256)
    ctxs = [Context(dim=256, pixscale=ps, precision='mixed', device=0) for _ in range(3)]
    try:
        one = ctxs[0].reconstruct(lb, see, gl, l0, three, (100, 10000))
        multi = Context.reconstruct_multi(ctxs, lb, see, gl, l0, three, (100, 10000))
        np.testing.assert_array_equal(multi['psf'], one['psf'])
        np.testing.assert_array_equal(multi['fit'], one['fit'])
        np.testing.assert_allclose(multi['psf_sum'], one['psf_sum'], rtol=1e-13)
        few = Context.reconstruct_multi(ctxs, lb, see[:2], gl[:2], l0[:2], three[:2], (100, 10000))
        np.testing.assert_array_equal(few['fit'], one['fit'][:2])
        with pytest.raises(MpsfrError) as e:       # 100 nm needs a crop far beyond the grid
            Context.reconstruct_multi(ctxs, np.array([100.0, 700.0]), see, gl, l0, three, (100, 10000))
        assert 'context' in str(e.value) and 'crop' in str(e.value)
        # contexts that would give an inconsistent table are refused: another precision, another pixel scale
        odd = [Context(dim=256, pixscale=ps, precision='f64', device=0), Context(dim=256, pixscale=1.1 * ps, device=0)]
        try:
            for o in odd:
                with pytest.raises(MpsfrError) as e:
                    Context.reconstruct_multi([ctxs[0], o], lb, see, gl, l0, three, (100, 10000))
                assert 'share' in str(e.value)
        finally:
            for o in odd:
                o.close()
    finally:
        for c in ctxs:
            c.close()


def test_asynchronous_multi_context_call(api):
    """mpsfr_reconstruct_multi_async / mpsfr_wait_multi: the shards of a table as asynchronous host-output calls on
    several contexts (here three on the one GPU), no host thread, the caller free until it waits -- against the
    blocking multi-context call bit for bit; one call pending per ctxs[0]; a failing shard leaves the contexts
    usable and the arrays of the shards already queued unwritten."""
    from muse_psfr_amd._lib import Context, MpsfrError
    see, gl, l0 = api.synthetic_rows(23)
    three = (np.arange(23) % 4 == 0).astype(np.uint8)
    lb = np.linspace(500, 900, 5)
    ps = api.grid_pixscale(256)
    ctxs = [Context(dim=256, pixscale=ps, precision='mixed', device=0) for _ in range(3)]
    try:
        want = Context.reconstruct_multi(ctxs, lb, see, gl, l0, three, (100, 10000))
        for _ in range(2):
            p = Context.reconstruct_multi_async(ctxs, lb, see, gl, l0, three, (100, 10000))
            with pytest.raises(MpsfrError):            # one pending call per ctxs[0]
                Context.reconstruct_multi_async(ctxs, lb, see, gl, l0, three, (100, 10000))
            got = p.wait()
            for k in ('psf', 'fit', 'psf_sum'):
                np.testing.assert_array_equal(got[k], want[k], err_msg=k)
            assert all(not c._pending for c in ctxs)
        few = Context.reconstruct_multi_async(ctxs, lb, see[:2], gl[:2], l0[:2], three[:2], (100, 10000)).wait()
        np.testing.assert_array_equal(few['fit'], want['fit'][:2])
        with pytest.raises(MpsfrError) as e:           # 100 nm: the first shard already fails
            Context.reconstruct_multi_async(ctxs, np.array([100.0, 700.0]), see, gl, l0, three, (100, 10000))
        assert 'context 0' in str(e.value)
        again = Context.reconstruct_multi_async(ctxs, lb, see, gl, l0, three, (100, 10000)).wait()
        np.testing.assert_array_equal(again['fit'], want['fit'])
        # a single-context asynchronous call still pending on a LATER context when a multi-context call
        # fails -- the failure abandons every context, so its wait() raises instead of handing back arrays nobody
        # wrote, nothing is left pending on either side, and the contexts stay usable
        single = ctxs[2].reconstruct_async(lb, see[:3], gl[:3], l0[:3], three[:3], (100, 10000))
        with pytest.raises(MpsfrError):
            Context.reconstruct_multi_async(ctxs, np.array([100.0, 700.0]), see, gl, l0, three, (100, 10000))
        assert all(not c._pending for c in ctxs)
        with pytest.raises(MpsfrError, match='abandoned'):
            single.wait()
        again = Context.reconstruct_multi_async(ctxs, lb, see, gl, l0, three, (100, 10000)).wait()
        np.testing.assert_array_equal(again['fit'], want['fit'])
    finally:
        for c in ctxs:
            c.close()


def test_asynchronous_host_outputs_equal_the_blocking_call(api):
    """on_device = 2 (Context.reconstruct_async): several calls in flight -- more than the ring of four
    staging sets holds, so that the library hands the oldest over by itself -- give, bit for bit, what
    the blocking call gives; tickets complete in order; mpsfr_sync hands everything over; a ticket that
    was never issued is an error; a failing call leaves the ring as it found it."""
    from muse_psfr_amd._lib import Context, MpsfrError
    lb = np.linspace(480, 920, 7)
    ps = api.grid_pixscale(256)
    sets = [api.synthetic_rows(9 + 3 * k, seed=100 + k) for k in range(7)]
    ctx = Context(dim=256, pixscale=ps, precision='mixed', device=0)
    try:
        want = [ctx.reconstruct(lb, s, g, l, None, (100, 10000)) for (s, g, l) in sets]
        pend = [ctx.reconstruct_async(lb, s, g, l, None, (100, 10000)) for (s, g, l) in sets]
        assert [p.ticket for p in pend] == list(range(7))
        got_last = pend[-1].wait()              # completes every earlier ticket too
        for p, w in zip(pend, want):
            r = p.wait()
            for k in ('psf', 'psf_sum', 'fit'):
                np.testing.assert_array_equal(r[k], w[k])
        np.testing.assert_array_equal(got_last['fit'], want[-1]['fit'])
        # mpsfr_sync hands the pending results over
        p1 = ctx.reconstruct_async(lb, *sets[0], None, (100, 10000), want_psf=False)
        p2 = ctx.reconstruct_async(lb, *sets[1], None, (100, 10000), want_psf=False)
        ctx.sync()
        np.testing.assert_array_equal(p1._arrays['fit'], want[0]['fit'])
        np.testing.assert_array_equal(p2._arrays['psf_sum'], want[1]['psf_sum'])
        assert p2._arrays['psf'] is None
        from muse_psfr_amd._lib import _check
        with pytest.raises(MpsfrError):
            _check(ctx.lib.mpsfr_wait(ctx._h, 99))
        with pytest.raises(MpsfrError):          # grid too small for 100 nm: the call fails ...
            ctx.reconstruct_async(np.array([100.0]), *sets[0], None, (100, 10000))
        p3 = ctx.reconstruct_async(lb, *sets[2], None, (100, 10000))     # ... and the next ticket is the next number
        assert p3.ticket == p2.ticket + 1
        np.testing.assert_array_equal(p3.wait()['fit'], want[2]['fit'])
    finally:
        ctx.close()


@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_stage_level_functions_against_the_reference_goldens(api, golden, ref_masks, prec):
    """simul_psd_wfm / psf_muse / convolve_final_psf of the reference (psfrec.py:36, 644, 874) as entry
    points of their own, on the reference's grid: the PSD against the fixtures G2 holds of it (centre
    block, two rows, total: captured from the real reference), then chained -- PSD -> stamps -> final
    stamps -- against G2's pre_* / fin_* stamps, one direction and nine."""
    g = golden('g2_native1280')
    lb = g['lbda']
    tol = dict(f64=1e-9, mixed=2e-5)[prec]
    for k in (0, 1, 4):
        see, gl_, l0_, npl, three = g['meta'][k]
        npl = int(npl)
        psd = api.simul_psd_wfm([gl_, 1 - gl_], H, see, l0_, npsflin=npl, three_lgs_mode=bool(three),
                                cutoff_masks=ref_masks, precision=prec, verbose=False)
        assert psd.shape == (npl * npl, 1280, 1280)
        c = 640
        assert rel_err(psd[:, c - 48:c + 48, c - 48:c + 48], g['psd_centre_%d' % k]) < 1e-12
        assert rel_err(psd[:, 0, :], g['psd_row0_%d' % k]) < 1e-12
        assert rel_err(psd[:, c, :], g['psd_rowc_%d' % k]) < 1e-12
        np.testing.assert_allclose(psd.sum(axis=(1, 2)), g['psd_sum_%d' % k], rtol=1e-11)
        pre = api.psf_muse(psd if npl > 1 else psd[0], lb, precision=prec)
        assert rel_err(pre, g['pre_%d' % k]) < tol
        fin = api.convolve_final_psf(lb, see, gl_, l0_, pre, precision=prec)
        assert rel_err(fin, g['fin_%d' % k]) < tol
        # the convolution alone, on the reference's own stamps
        fin2 = api.convolve_final_psf(lb, see, gl_, l0_, g['pre_%d' % k], precision=prec)
        assert rel_err(fin2, g['fin_%d' % k]) < (1e-10 if prec == 'f64' else 2e-6)
        record_margin('stage_functions_%s' % prec, stamp_pre=rel_err(pre, g['pre_%d' % k]), stamp=rel_err(fin, g['fin_%d' % k]))


def test_psf_muse_takes_any_psd(api):
    """psf_muse on a PSD that is NOT the model's (no symmetry between the rows su and -1-su, a tilted
    ridge, another zenith angle): against the oracle's reference-shaped psf_stamps on a 256^2 grid."""
    dim, ps = 256, api.grid_pixscale(256)
    lb = np.array([480.0, 700.0, 930.0])
    tabs = O.ao_tables(H, False, 1, exact_masks=True)
    psd = O.residual_psd([0.6, 0.4], H, 0.9, 18.0, 1, dim, False, tables=tabs)[0]
    yy, xx = np.mgrid[:dim, :dim] - dim // 2
    psd = psd * (1.0 + 0.3 * np.exp(-((xx - 7) ** 2 + (yy + 3) ** 2) / 50.0) + 0.2 * (np.hypot(xx, yy) > 20) * (xx > 2 * yy))
    want = O.psf_stamps_refshaped(psd[None], lb, 40, ps)
    for prec, tol in (('f64', 1e-9), ('mixed', 2e-5)):
        got = api.psf_muse(psd, lb, pixscale=ps, precision=prec)
        assert rel_err(got, want) < tol, prec
    # the zenith angle only rescales r0 (psfrec.py:108, 183-187)
    a = api.simul_psd_wfm([0.6, 0.4], H, 0.9, 18.0, zenith=30.0, dim=256, cutoff_masks='exact', verbose=False)
    b = O.residual_psd([0.6, 0.4], H, 0.9 / np.cos(np.deg2rad(30.0)) ** 0.6, 18.0, 1, 256, False, tables=tabs)
    assert rel_err(a, b) < 1e-12


def test_large_table_in_asynchronous_parts_equals_one_call(api, monkeypatch):
    """compute_psf_from_sparta on a table of >= 250 tasks runs as asynchronous parts with the FIT_ROWS
    records assembled part by part (psfrec._reconstruct_pipelined): same table, bit for bit, as the
    single blocking call; PSF_MEAN to rounding (the parts' stamp sums are added in order)."""
    from muse_psfr_amd import psfrec
    n = 300
    see, gl, l0 = api.synthetic_rows(n)
    tbl = api.create_sparta_table(nlines=n)
    for k in range(1, 5):
        tbl.data['LGS%d_SEEING' % k][:] = see
        tbl.data['LGS%d_TUR_GND' % k][:] = gl
        tbl.data['LGS%d_L0' % k][:] = l0
    tbl.data['LGS2_L0'][::7] = 150.0             # three-laser rows
    kw = dict(verbose=False, dim=256, pixscale=api.grid_pixscale(256), lmin=500, lmax=900, nl=5, mean_of_lgs=False)
    a = api.compute_psf_from_sparta(_hdul(tbl), **kw)
    monkeypatch.setattr(psfrec, 'PIPELINE_MIN_TASKS', 10 ** 9)
    b = api.compute_psf_from_sparta(_hdul(tbl), **kw)
    ra, rb = a['FIT_ROWS'].data, b['FIT_ROWS'].data
    assert ra.dtype == rb.dtype and ra.shape == rb.shape == (1200 * 5 - 5 * len(range(0, n, 7)),)
    for name in ra.dtype.names:
        np.testing.assert_array_equal(ra[name], rb[name], err_msg=name)
    np.testing.assert_allclose(a['PSF_MEAN'].data, b['PSF_MEAN'].data, rtol=1e-12)
    for name in a['FIT_MEAN'].data.dtype.names:
        np.testing.assert_allclose(a['FIT_MEAN'].data[name], b['FIT_MEAN'].data[name], rtol=1e-9, atol=1e-12)


def test_large_table_on_a_grid_too_small_raises_like_the_reference(api):
    """The part-wise path keeps the error behaviour of the single call: a wavelength whose stamp field does not
    fit the grid is a ValueError (the reference fails in scipy's interpn, psfrec.py:663-683), and the context
    stays usable."""
    n = 260
    see, gl, l0 = api.synthetic_rows(n)
    tbl = api.create_sparta_table(nlines=n)
    for k in range(1, 5):
        tbl.data['LGS%d_SEEING' % k][:] = see
        tbl.data['LGS%d_TUR_GND' % k][:] = gl
        tbl.data['LGS%d_L0' % k][:] = l0
    kw = dict(verbose=False, dim=128, pixscale=api.grid_pixscale(128), nl=3)
    with pytest.raises(ValueError):
        api.compute_psf_from_sparta(_hdul(tbl), lmin=200, lmax=300, **kw)
    res = api.compute_psf_from_sparta(_hdul(tbl), lmin=500, lmax=900, **kw)
    assert len(res['FIT_ROWS'].data) == n * 3


def _big_table(api, n):
    see, gl, l0 = api.synthetic_rows(n)
    tbl = api.create_sparta_table(nlines=n)
    for k in range(1, 5):
        tbl.data['LGS%d_SEEING' % k][:] = see
        tbl.data['LGS%d_TUR_GND' % k][:] = gl
        tbl.data['LGS%d_L0' % k][:] = l0
    return tbl


@pytest.mark.parametrize('exc', [RuntimeError, KeyboardInterrupt])
def test_large_table_whose_later_part_fails_leaves_no_pending_arrays(api, monkeypatch, exc):
    """a failure AFTER the first asynchronous parts have been queued (any exception, not only a
    library error).  The queued parts are abandoned -- mpsfr_abandon: the GPU drains, the library forgets the
    output arrays without writing to them -- so the cached context holds no pointer into freed memory, and the
    next table through the same context is right."""
    import gc
    from muse_psfr_amd import psfrec
    from muse_psfr_amd._lib import Context
    monkeypatch.setattr(psfrec, 'PIPELINE_PARTS', 4)
    tbl = _big_table(api, 520)                    # four parts
    kw = dict(verbose=False, dim=128, pixscale=api.grid_pixscale(128), lmin=500, lmax=900, nl=3)
    want = api.compute_psf_from_sparta(_hdul(tbl), **kw)
    real, calls, seen = Context.reconstruct_async, [], []

    def failing(self, *a, **k):
        calls.append(1)
        if len(calls) == 3:
            seen.append((self, dict(self._pending)))
            raise exc('part 3')
        return real(self, *a, **k)
    monkeypatch.setattr(Context, 'reconstruct_async', failing)
    with pytest.raises(exc):
        api.compute_psf_from_sparta(_hdul(tbl), **kw)
    monkeypatch.setattr(Context, 'reconstruct_async', real)
    ctx, held = seen[0]
    assert len(held) == 2 and not ctx._pending and ctx._abandoned >= set(held)
    # the two queued parts were never handed over: their arrays still hold what np.empty left (the library
    # did not write them), and waiting for one of them now says so
    gc.collect()
    got = api.compute_psf_from_sparta(_hdul(tbl), **kw)
    for name in want['FIT_ROWS'].data.dtype.names:
        np.testing.assert_array_equal(want['FIT_ROWS'].data[name], got['FIT_ROWS'].data[name], err_msg=name)
    np.testing.assert_array_equal(want['PSF_MEAN'].data, got['PSF_MEAN'].data)


def test_dropped_pending_results_are_kept_alive_by_the_context(api):
    """Context.reconstruct_async results that are dropped without wait().  The library holds raw
    pointers to their arrays until the ticket is handed over (in a wait, a sync, or when the ring of four comes
    round): the context keeps the arrays alive for exactly that long."""
    import gc
    import weakref
    ps = api.grid_pixscale(128)
    ctx = api.Context(dim=128, pixscale=ps, precision='mixed')
    see, gl, l0 = api.synthetic_rows(6)
    lb = np.array([500.0, 700.0, 900.0])
    want = ctx.reconstruct(lb, see, gl, l0, None, H)
    refs = []
    for k in range(9):                               # more than twice round the ring, nobody waits
        p = ctx.reconstruct_async(lb, see, gl, l0, None, H)
        refs.append(weakref.ref(p._arrays['fit']))
        del p
        gc.collect()
        assert len(ctx._pending) <= 4 and refs[-1]() is not None
        if k >= 4:
            assert refs[k - 4]() is None                 # handed over by the call four tickets later: released
    last = ctx.reconstruct_async(lb, see, gl, l0, None, H)
    r = last.wait()
    assert not ctx._pending and all(w() is None for w in refs)
    assert np.array_equal(r['fit'], want['fit']) and np.array_equal(r['psf'], want['psf'])
    # abandon: the arrays of the calls in flight are never written
    p = ctx.reconstruct_async(lb, see, gl, l0, None, H)
    p._arrays['fit'][:] = -7.0
    ctx.abandon()
    ctx.sync()
    assert np.all(p._arrays['fit'] == -7.0)
    with pytest.raises(api.MpsfrError):
        p.wait()
    r = ctx.reconstruct_async(lb, see, gl, l0, None, H).wait()      # the context stays usable
    assert np.array_equal(r['fit'], want['fit'])
    ctx.close()
