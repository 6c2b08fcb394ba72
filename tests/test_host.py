"""CPU: host-side logic of the drop-in layer (no GPU compute)."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import psfr_oracle as O
from conftest import ROOT


def test_synthetic_rows_are_a_stable_stream():
    from muse_psfr_amd import synthetic_rows
    s, g, l = synthetic_rows(1000)
    s2, g2, l2 = synthetic_rows(100)
    assert np.array_equal(s[:100], s2) and np.array_equal(g[:100], g2) and np.array_equal(l[:100], l2)
    assert s.min() >= 0.4 and s.max() <= 1.6 and g.min() >= 0.30 and g.max() <= 0.95
    assert l.min() > 8 and l.max() < 30                      # validity window psfrec.py:1049-1051
    np.testing.assert_allclose([s[0], g[0], l[0]], [0.651657175502, 0.746617995631, 23.559763217051],
                               rtol=1e-9)


def test_grid_pixscale_rule():
    from muse_psfr_amd import grid_pixscale
    assert grid_pixscale(1280) == 0.2
    for n in (128, 256, 512, 1024):          # npixc(465 nm) <= N
        assert O.npix_crop(np.array([465.0]), 40, grid_pixscale(n))[0] <= n


def test_host_masks_follow_this_numpy_like_the_oracle():
    from muse_psfr_amd import host_cutoff_masks
    a = host_cutoff_masks()
    b = O.numpy_cutoff_masks()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_minifits_roundtrip_and_sparta_table(tmp_path):
    from muse_psfr_amd import _minifits as mf
    from muse_psfr_amd import create_sparta_table
    hdu = create_sparta_table(nlines=3, seeing=0.9, L0=21, GL=0.6, bad_l0=True)
    assert hdu.name == 'SPARTA_ATM_DATA'
    d = np.array(hdu.data)
    assert len(d) == 3 and list(d.dtype.names)[:3] == ['LGS1_SEEING', 'LGS1_TUR_GND', 'LGS1_L0']
    assert d['LGS4_L0'][0] == 150 and d['LGS3_L0'][0] == 21
    if isinstance(hdu, mf.BinTableHDU):
        path = str(tmp_path / 's.fits')
        img = mf.ImageHDU(np.arange(24, dtype=float).reshape(2, 3, 4), name='IMG')
        t2 = mf.BinTableHDU.from_columns({'a': np.arange(3), 'v': np.ones((3, 2)), 'f': np.array([True, False, True])},
                                         name='T2')
        t2.header['SEEING'] = 1.25
        mf.HDUList([mf.PrimaryHDU(), hdu, img, t2]).writeto(path, overwrite=True)
        assert os.path.getsize(path) % 2880 == 0
        back = mf.open(path)
        assert [h.name for h in back] == ['PRIMARY', 'SPARTA_ATM_DATA', 'IMG', 'T2']
        assert np.array_equal(back['SPARTA_ATM_DATA'].data['LGS2_SEEING'], d['LGS2_SEEING'])
        assert np.array_equal(back['IMG'].data, img.data)
        assert np.array_equal(back['T2'].data['v'], np.ones((3, 2))) and back['T2'].header['SEEING'] == 1.25
        assert list(back['T2'].data['f']) == [True, False, True]
        with pytest.raises(OSError):
            mf.HDUList([mf.PrimaryHDU()]).writeto(path)
        # row-field assignment as the reference's tests do (test_psfrec.py:50-55)
        back['SPARTA_ATM_DATA'].data[0]['LGS1_L0'] = 20
        assert back['SPARTA_ATM_DATA'].data['LGS1_L0'][0] == 20
        # file-object round trip (cli.py:38-43 writes the table to a BytesIO)
        bio = io.BytesIO()
        create_sparta_table(outfile=bio)
        bio.seek(0)
        assert mf.open(bio)['SPARTA_ATM_DATA'].data['LGS1_L0'][0] == 25


def test_minifits_files_are_read_by_astropy(tmp_path):
    """Files written by the NumPy FITS writer are standard: astropy (conda interpreter) reads them."""
    py39 = '/opt/conda/bin/python3.9'
    if not os.path.exists(py39):
        pytest.skip('no astropy interpreter')
    from muse_psfr_amd import _minifits as mf
    path = str(tmp_path / 'x.fits')
    t = mf.BinTableHDU.from_columns({'lbda': np.array([500., 700.]), 'fwhm': np.array([[1., 1.], [2., 2.]]),
                                     'row_idx': np.array([1, 2])}, name='FIT_ROWS')
    t.header['L0'] = 25.0
    mf.HDUList([mf.PrimaryHDU(), t, mf.ImageHDU(np.ones((2, 4, 4)), name='PSF_MEAN')]).writeto(path)
    code = ("import numpy as np\nnp.asscalar=lambda a:a.item(); np.alen=len\n"
            "from astropy.io import fits\nh=fits.open(%r)\n"
            "assert [x.name for x in h]==['PRIMARY','FIT_ROWS','PSF_MEAN']\n"
            "assert h['FIT_ROWS'].data['fwhm'].shape==(2,2) and h['FIT_ROWS'].header['L0']==25.0\n"
            "assert h['FIT_ROWS'].data['row_idx'][1]==2 and h['PSF_MEAN'].data.shape==(2,4,4)\nprint('ok')" % path)
    r = subprocess.run([py39, '-B', '-W', 'ignore', '-c', code], capture_output=True, text=True)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


def test_sparta_front_end_row_filtering(monkeypatch, caplog):
    """Outlier rejection, 3-LGS flag, mean_of_lgs bookkeeping and the exact log strings of the
    reference (psfrec.py:1041-1080; test_psfrec.py:79-81, 99-100) -- GPU call stubbed out."""
    import logging
    from muse_psfr_amd import psfrec, _minifits as mf
    calls = {}

    def fake(lbda, tasks, npsflin, h, dim, dimpsf, pixscale, precision, cutoff_masks, device, want_psf=True,
             devices=None, n_jobs=1):
        stats, three = tasks            # ([ntask][3] seeing / GL / L0, [ntask] three-laser mode)
        calls['tasks'] = [tuple(float(x) for x in st) + (bool(t3),) for st, t3 in zip(stats, three)]
        n, nl = len(stats), len(lbda)
        fit = np.zeros((n, nl, 16))
        fit[:, :, 3] = 2.0
        fit[:, :, 4] = 2.5
        fit[:, :, 0] = 1.0
        return dict(psf=None, psf_sum=np.ones((nl, 40, 40)) * n, fit=fit)

    class Ctx:
        def fit_stamps(self, st):
            f = np.zeros((len(st), 16)); f[:, 0] = 1; f[:, 3] = 2; f[:, 4] = 2.5
            return f
    monkeypatch.setattr(psfrec, '_reconstruct', fake)
    monkeypatch.setattr(psfrec, 'get_context', lambda *a, **k: Ctx())
    monkeypatch.setattr(psfrec, '_astropy', lambda: (None, None))
    tbl = psfrec.create_sparta_table(nlines=2)
    tbl.data[0]['LGS1_L0'] = 20
    tbl.data[0]['LGS1_SEEING'] = 0.8
    tbl.data[0]['LGS1_TUR_GND'] = 0.5
    tbl.data[0]['LGS3_L0'] = 100           # rejected (> MAX_L0)
    tbl.data[1]['LGS2_TUR_GND'] = 0.0      # rejected (GL > 0 fails)
    tbl.data[1]['LGS4_L0'] = 5             # rejected (< MIN_L0)
    hdul = mf.HDUList([mf.PrimaryHDU(), tbl])
    with caplog.at_level(logging.INFO, logger='muse_psfr_amd.psfrec'):
        res = psfrec.compute_psf_from_sparta(hdul, lmin=500, lmax=700, nl=3, mean_of_lgs=False)
    msgs = [r.message for r in caplog.records]
    assert msgs[0] == 'Processing SPARTA table with 2 values, njobs=-1 ...'
    assert msgs[1] == '1/2 : Using only 3 values out of 4 after outliers rejection'
    assert msgs[2] == '2/2 : Using only 2 values out of 4 after outliers rejection'
    assert 'Using three lasers mode' in msgs
    assert [t[3] for t in calls['tasks']] == [True] * 5
    assert calls['tasks'][0][:3] == (0.8, 0.5, 20.0)
    assert [h.name for h in res] == ['PRIMARY', 'SPARTA_ATM_DATA', 'FIT_ROWS', 'FIT_MEAN', 'PSF_MEAN']
    rows = res['FIT_ROWS'].data
    assert list(rows['lgs_idx'][::3]) == [1, 2, 4, 1, 3] and list(rows['row_idx'][::3]) == [1, 2, 3, 4, 5]
    assert res['FIT_MEAN'].header['L0'] == 25.0 and res['PSF_MEAN'].data.shape == (3, 40, 40)
    assert 'SEEING' not in res['FIT_ROWS'].header
    # mean_of_lgs=True averages the valid lasers
    caplog.clear()
    psfrec.compute_psf_from_sparta(hdul, lmin=500, lmax=700, nl=3)
    np.testing.assert_allclose(calls['tasks'][0][:3], ((0.8 + 1 + 1) / 3, (0.5 + .7 + .7) / 3, (20 + 25 + 25) / 3))
    # nothing valid -> None + the reference's messages (test_psfrec.py:99-100)
    caplog.clear()
    bad = mf.HDUList([mf.PrimaryHDU(), psfrec.create_sparta_table(L0=1000)])
    with caplog.at_level(logging.INFO, logger='muse_psfr_amd.psfrec'):
        assert psfrec.compute_psf_from_sparta(bad) is None
    assert caplog.records[1].message == '1/1 : No valid values, skipping this row'
    assert caplog.records[2].message == 'No valid values'
    with pytest.raises(KeyError):
        psfrec.compute_psf_from_sparta(hdul, extname='NOPE')


def test_fit_poly_and_intrinsic_psf():
    from muse_psfr_amd import fit_psf_with_polynom, muse_intrinsic_psf
    lb = np.linspace(500, 900, 9)
    fwhm, beta, _, _ = muse_intrinsic_psf(lb)
    of, ob = O.muse_intrinsic_psf(lb)
    assert np.array_equal(fwhm, of) and np.array_equal(beta, ob)
    r = fit_psf_with_polynom(lb, fwhm, beta, deg=(5, 5), output=1)
    assert r['fwhm_fit'].shape == (50,) and len(r['fwhm_pol']) == 6


def test_shard_bounds():
    from muse_psfr_amd.distributed import shard_bounds
    assert shard_bounds(1000, 8) == [(i * 125, (i + 1) * 125) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_cli_argument_errors():
    """test_psfrec.py:104-108: the CLI's argument errors (no GPU needed)."""
    from muse_psfr_amd.cli import main
    with pytest.raises(SystemExit, match='no input file provided'):
        main([])
    with pytest.raises(SystemExit, match='--values must contain a list.*'):
        main(['--values', '0.1,0.2'])


def test_direction_perf_and_radial_profile():
    """psfrec.py:154-158 (n=1 -> (0,0); n=2 -> {-30,0}^2; n=3 -> {-30,0,30}^2) and
    psfrec.py:810-823."""
    from muse_psfr_amd import direction_perf, radial_profile
    assert direction_perf(1).tolist() == [[0.0], [0.0]]
    d2 = direction_perf(2)
    assert sorted(set(d2[0])) == [-30.0, 0.0] and d2.shape == (2, 4)
    d3 = direction_perf(3)
    assert d3.shape == (2, 9) and sorted(set(d3[1])) == [-30.0, 0.0, 30.0]
    assert d3[:, 0].tolist() == [-30.0, -30.0] and d3[:, 1].tolist() == [-30.0, 0.0]   # y fastest
    np.testing.assert_array_equal(direction_perf(3), O.eval_directions(3))
    # a radially symmetric image comes back as its own profile
    n = 40
    x, y = np.ogrid[:n, :n]
    r = np.hypot(x - 20, y - 20)
    centers, prof = radial_profile(np.ones((n, n)))
    assert centers[0] == 0.5 and np.allclose(np.diff(centers), 1.0)
    assert np.allclose(prof[np.isfinite(prof)], 1.0)
    centers, prof = radial_profile(np.floor(r))           # ring k holds the value k exactly
    k = np.arange(len(prof))
    ok = np.isfinite(prof)
    np.testing.assert_allclose(prof[ok], k[ok])
    assert len(prof) == int(np.round(r.max()) + 1)
    c2, p2 = radial_profile(np.floor(r), binsize=2)
    assert c2[0] == 1.0 and len(p2) == int(np.round(r.max() / 2) + 1)


def test_plot_psf_on_a_result_list(tmp_path):
    """plot_psf (psfrec.py:826-858) on an HDUList and on the file written from it."""
    import matplotlib
    matplotlib.use('agg', force=True)
    from muse_psfr_amd import _minifits as mf
    from muse_psfr_amd import plot_psf
    from muse_psfr_amd.psfrec import _table_hdu
    lb = np.linspace(490, 930, 5)
    x, y = np.ogrid[:40, :40]
    psf = np.array([(1 + ((x - 20) ** 2 + (y - 20) ** 2) / (4.0 + k)) ** -2.5 for k in range(5)])
    fit = _table_hdu({'lbda': lb, 'fwhm': np.repeat(np.linspace(0.8, 0.6, 5)[:, None], 2, axis=1),
                      'n': np.linspace(2.7, 2.2, 5)}, {'SEEING': 1.0}, 'FIT_MEAN')
    if isinstance(fit, mf.BinTableHDU):
        hdul = mf.HDUList([mf.PrimaryHDU(), fit, mf.ImageHDU(data=psf, name='PSF_MEAN')])
    else:
        from astropy.io import fits
        hdul = fits.HDUList([fits.PrimaryHDU(), fit, fits.ImageHDU(data=psf, name='PSF_MEAN')])
    fig = plot_psf(hdul, npsflin=3)
    assert len(fig.axes) == 7                    # 2 x 3 panels + the colour bar
    titles = [a.get_title() for a in fig.axes]
    assert 'PSF' in titles and 'radial profile' in titles
    xs, ys = fig.axes[4].lines[0].get_data()     # FWHM(lambda) panel
    np.testing.assert_allclose(xs, lb)
    np.testing.assert_allclose(ys, np.linspace(0.8, 0.6, 5))
    assert len(fig.axes[2].collections) == 2     # directions + LGS
    fig.savefig(str(tmp_path / 'fig.png'))
    out = str(tmp_path / 'res.fits')
    hdul.writeto(out, overwrite=True)
    fig2 = plot_psf(out)
    fig2.savefig(str(tmp_path / 'fig2.png'))
    assert os.path.getsize(str(tmp_path / 'fig2.png')) > 0


def test_fanout_devices(monkeypatch):
    """The devices a table runs on (psfrec._fanout_devices): an explicit device is that device and no
    other; the automatic choice fans out over the visible GPUs for large tables only, is capped by
    n_jobs, and never leaves its own GPU inside a one-rank-per-GPU launch."""
    import muse_psfr_amd._lib as L
    from muse_psfr_amd import psfrec
    monkeypatch.setattr(L, 'device_count', lambda: 8)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('LOCAL_RANK', raising=False)
    f = psfrec._fanout_devices
    assert f(None, 3, 1000, -1) == [3]                 # explicit device: nothing else, whatever n_jobs
    assert f(None, 3, 1000, 1) == [3]
    assert f(None, 0, 100000, 4) == [0]
    assert f(None, None, 1000, -1) == list(range(8))   # automatic: every visible GPU
    assert f(None, None, 1000, 2) == [0, 1]            # capped by n_jobs
    assert f(None, None, 1000, 1) == [0]
    assert f(None, None, 100, -1) == [0, 1, 2]         # 32 tasks per device at least
    assert f(None, None, 40, -1) == [0]
    assert f([2, 2, 5], None, 3, 1) == [2, 2, 5]       # an explicit list is taken as given
    with pytest.raises(ValueError):
        f([], None, 10, -1)
    monkeypatch.setenv('WORLD_SIZE', '8')              # a rank of torch.distributed.run stays at home
    monkeypatch.setenv('LOCAL_RANK', '5')
    assert f(None, None, 100000, -1) == [5]
    assert f(None, 2, 100000, -1) == [2]
    monkeypatch.setattr(L, 'device_count', lambda: 1)
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.delenv('LOCAL_RANK')
    assert f(None, None, 100000, -1) == [0]


def test_fit_rows_records_equal_the_column_wise_table():
    """The FIT_ROWS table assembled row-wise in one structured array (compute_psf_from_sparta without
    astropy) holds the same dtype and values as the column-by-column form (psfrec.py:1086-1101)."""
    from muse_psfr_amd import psfrec as P, _minifits as mf
    rng = np.random.default_rng(4)
    nt, nl = 9, 6
    fit = rng.random((nt, nl, 16)) + 0.5
    fit[:, :, 4] += 1.5
    fit[2, 3, 4] = 1.0                      # n = 1: the relative error of the flux divides by n - 1
    lb = np.linspace(490, 930, nl)
    stats, las = rng.random((nt, 3)), np.arange(nt) % 4 + 1
    rec = P._fit_rows_records(lb, fit, 0.2, stats, las)
    part, blk = P._fit_rows_template(lb, stats, las)          # filled in two parts, as the asynchronous path does
    P._fit_rows_fill(blk[:4 * nl], fit[:4], 0.2)
    P._fit_rows_fill(blk[4 * nl:], fit[4:], 0.2)
    cols = P._fit_columns(np.tile(lb, nt), fit.reshape(nt * nl, -1), 0.2)
    for j, k in enumerate(('SEEING', 'GL', 'L0')):
        cols[k] = np.repeat(stats[:, j], nl)
    cols['row_idx'] = np.repeat(np.arange(1, nt + 1), nl)
    cols['lgs_idx'] = np.repeat(las, nl)
    ref = mf.BinTableHDU.from_columns(cols, mf.Header(), 'FIT_ROWS').data
    assert rec.dtype == ref.dtype and rec.shape == ref.shape
    for name in ref.dtype.names:
        np.testing.assert_array_equal(rec[name], ref[name])
        np.testing.assert_array_equal(part[name], ref[name])
