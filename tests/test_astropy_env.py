"""CPU: the astropy branches of the drop-in layer, run ONCE in the reference's own environment.

`compute_psf` returns an astropy Table (psfrec.py:974-978) and `compute_psf_from_sparta` an astropy HDUList
(:1086-1113).  The interpreter the tests run under has no astropy (neither has the GPU box), so those branches of
muse_psfr_amd/psfrec.py are exercised in a subprocess under the one interpreter of the build container that has it
(/opt/conda/bin/python3.9, astropy 4.3.1 -- the interpreter oracle/make_golden.py ran the real reference under),
with the GPU context replaced by a stub that answers from tests/golden/g7_sparta_lgs.npz (tests/astropy_env_driver.py).
What comes out is compared with what the REFERENCE's compute_psf_from_sparta returned for the same table (G7):
HDU names and types, the FIT_ROWS / FIT_MEAN column names in the reference's order, dtypes, the (n, 2) shapes,
row_idx / lgs_idx, the header cards SEEING / GL / L0 present in FIT_MEAN and absent in FIT_ROWS (:1095-1112), values.

Which interpreter covers which branch: INTEGRATION.md, "Interpreters".  Skipped where that interpreter is absent.
"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = os.environ.get('MPSFR_ASTROPY_PYTHON', '/opt/conda/bin/python3.9')

# psfrec.py:866-870 (fit_psf_cube: the fit object's attributes less ima/rot/cont/err_rot/err_cont, 'lbda' in front),
# :975-977 (SEEING, GL, L0), :1091-1092 (row_idx, lgs_idx) -- asserted on the real reference's table by
# oracle/make_golden.py when G7 was written
REF_FIT_COLS = ['lbda', 'center', 'flux', 'fwhm', 'n', 'peak', 'err_center', 'err_flux', 'err_fwhm', 'err_n', 'err_peak']
REF_ROWS_COLS = REF_FIT_COLS + ['SEEING', 'GL', 'L0', 'row_idx', 'lgs_idx']


@pytest.fixture(scope='module')
def env_run():
    if not os.path.exists(PY):
        pytest.skip('no interpreter with astropy here (%s)' % PY)
    probe = subprocess.run([PY, '-c', 'import numpy as np\n'
                            'np.asscalar = getattr(np, "asscalar", None) or (lambda a: a.item())\n'
                            'np.alen = getattr(np, "alen", None) or len\nimport astropy'],
                           capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip('astropy does not import under %s' % PY)
    env = dict(os.environ)
    # (conda ships an older libstdc++ than the one libmpsfr.so was linked against: mpsfr_fit_rows -- pure host C,
    # used by the asynchronous-parts path -- needs the system's)
    sysstd = '/usr/lib/x86_64-linux-gnu/libstdc++.so.6'
    if os.path.exists(sysstd):
        env['LD_PRELOAD'] = sysstd
    env.pop('PYTHONPATH', None)
    p = subprocess.run([PY, '-W', 'ignore', os.path.join(ROOT, 'tests', 'astropy_env_driver.py'), ROOT],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout)


def test_astropy_is_really_used(env_run):
    assert env_run['astropy'] == [True, True]


def test_compute_psf_returns_an_astropy_table(env_run):
    r = env_run['compute_psf']
    assert r['type'] == 'astropy.table.table.Table'
    assert r['colnames'] == REF_FIT_COLS + ['SEEING', 'GL', 'L0']                 # psfrec.py:975-977
    assert r['meta'] == {'SEEING': 1.0, 'GL': 0.7, 'L0': 25.0}                    # :974
    assert r['len'] == 4 and r['cube_shape'] == [4, 40, 40]
    for c in ('center', 'fwhm', 'err_center', 'err_fwhm'):
        assert r['shapes'][c] == [4, 2], c
    assert all(v == 'float64' for v in r['dtypes'].values())


@pytest.mark.parametrize('path', ['one_call', 'async_parts'])
@pytest.mark.parametrize('tag', ['mean', 'lgs'])
def test_compute_psf_from_sparta_returns_what_the_reference_returns(env_run, golden, tag, path):
    g = golden('g7_sparta_lgs')
    r = env_run['%s_%s' % (tag, path)]
    assert r['type'] == 'astropy.io.fits.hdu.hdulist.HDUList'
    assert r['names'] == ['PRIMARY', 'SPARTA_ATM_DATA', 'FIT_ROWS', 'FIT_MEAN', 'PSF_MEAN']     # :1086-1113
    assert r['hdu_types'] == ['PrimaryHDU', 'BinTableHDU', 'BinTableHDU', 'BinTableHDU', 'ImageHDU']
    assert r['roundtrip_names'] == r['names']
    assert r['sparta_copy_rows'] == 5
    # the library was asked the way the path says
    kinds = [c[0] for c in r['calls']]
    assert kinds == (['reconstruct', 'fit_stamps'] if path == 'one_call' else
                     ['reconstruct_async', 'reconstruct_async', 'fit_stamps'])
    rows, mean = r['FIT_ROWS'], r['FIT_MEAN']
    nrows = len(g['%s_rows_lbda' % tag])
    assert r['roundtrip_rows'] == nrows
    # column names in the reference's order
    assert rows['colnames'] == REF_ROWS_COLS
    assert mean['colnames'] == REF_FIT_COLS
    # header cards (psfrec.py:1095-1098 removes them from FIT_ROWS; :1106-1109 puts the medians into FIT_MEAN)
    assert rows['header_has'] == {'SEEING': False, 'GL': False, 'L0': False}
    assert mean['header_has'] == {'SEEING': True, 'GL': True, 'L0': True}
    np.testing.assert_allclose([mean['header'][k] for k in ('SEEING', 'GL', 'L0')], g['%s_mean_hdr' % tag], rtol=1e-14)
    # dtypes and shapes
    for c in REF_ROWS_COLS:
        want = np.asarray(g['%s_rows_%s' % (tag, c)])
        assert rows['shapes'][c] == list(want.shape), c
        assert rows['dtypes'][c] == str(want.dtype), (c, rows['dtypes'][c], want.dtype)
    for c in REF_FIT_COLS:
        want = np.asarray(g['%s_mean_%s' % (tag, c)])
        assert mean['shapes'][c] == list(want.shape), c
        assert mean['dtypes'][c] == str(want.dtype), c
    # values: the stub's fit rows are the reference's columns turned inside out, so every column must come back
    # (err_flux through the quadrature formula of the layer: 1e-12; the others exactly)
    for c in REF_ROWS_COLS:
        np.testing.assert_allclose(np.asarray(rows['values'][c]), g['%s_rows_%s' % (tag, c)], rtol=1e-11, atol=0, err_msg=c)
    for c in REF_FIT_COLS:
        np.testing.assert_allclose(np.asarray(mean['values'][c]), g['%s_mean_%s' % (tag, c)], rtol=1e-11, atol=0, err_msg=c)
    assert np.asarray(rows['values']['row_idx']).tolist() == np.asarray(g['%s_rows_row_idx' % tag]).tolist()
    assert np.asarray(rows['values']['lgs_idx']).tolist() == np.asarray(g['%s_rows_lgs_idx' % tag]).tolist()
    assert r['PSF_MEAN']['shape'] == [4, 40, 40] and r['PSF_MEAN']['max_abs_diff'] < 1e-15


def test_create_sparta_table_is_an_astropy_hdu(env_run):
    r = env_run['create_sparta_table']
    assert r == {'type': 'BinTableHDU', 'name': 'SPARTA_ATM_DATA', 'ncols': 12, 'lgs4_l0': [150.0, 150.0, 150.0]}
