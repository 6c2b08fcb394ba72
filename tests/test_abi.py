"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mpsfr.h
declares (no compute calls -- there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'mpsfr.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mpsfr_[a-z_]+)\s*\(', src)))


def test_library_builds_and_exports_the_header():
    from muse_psfr_amd._build import build_library
    from muse_psfr_amd import _lib
    path = build_library(force=False, verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.mpsfr_version() == 102
    from muse_psfr_amd._build import source_hash
    assert lib.mpsfr_build_id().decode() == source_hash()
    assert lib.mpsfr_profile_count() == 16
    assert lib.mpsfr_profile_name(6) == b'otf_rowfft' and lib.mpsfr_profile_name(12) == b'otf_mfma'


def test_header_constants_match_python():
    from muse_psfr_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'mpsfr.h')).read()
    assert int(re.search(r'#define MPSFR_NFIT\s+(\d+)', src).group(1)) == _lib.NFIT
    assert int(re.search(r'#define MPSFR_DIM_AO\s+(\d+)', src).group(1)) == _lib.DIM_AO
    assert int(re.search(r'#define MPSFR_E_GRID\s+(-\d+)', src).group(1)) == _lib.E_GRID


def test_no_cpu_fallback_in_the_product():
    """The package must never import the oracle."""
    pkg = os.path.join(ROOT, 'muse_psfr_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert 'psfr_oracle' not in txt and 'import oracle' not in txt, f


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from muse_psfr_amd import Context, MpsfrError
    with pytest.raises(MpsfrError):
        Context(dim=128, pixscale=0.019)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/mpsfr.h is the boundary a non-Python caller binds to: it must compile as C99 (no C++
    in the signatures) and a C program must link against libmpsfr.so.  Without a GPU mpsfr_create
    fails loudly (an error code and a message), which is all this program checks."""
    import shutil
    import subprocess
    from muse_psfr_amd._build import build_library
    lib = build_library(force=False, verbose=False)
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'abi_c.c'
    src.write_text(
        '#include "mpsfr.h"\n'
        '#include <stdio.h>\n'
        'int main(void) {\n'
        '    mpsfr_ctx* c = 0;\n'
        '    int rc = mpsfr_create(&c, 0, 512, 40, 0.0762, 0);\n'
        '    if (rc == 0) { mpsfr_destroy(c); puts("created"); return 0; }\n'
        '    printf("rc=%d msg=%s version=%d devices=%d\\n", rc, mpsfr_last_error(), mpsfr_version(),\n'
        '           mpsfr_device_count());\n'
        '    return mpsfr_last_error()[0] ? 0 : 1;\n'
        '}\n')
    exe = tmp_path / 'abi_c'
    libdir = os.path.dirname(lib)
    subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'),
                    str(src), '-L', libdir, '-lmpsfr', '-Wl,-rpath,' + libdir, '-o', str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'created' in r.stdout or 'rc=-' in r.stdout


def test_fit_rows_helper_is_the_numpy_formula():
    """mpsfr_fit_rows (pure host C, no GPU): the 14 fit columns of FIT_ROWS (psfrec.py:866-870) from library fit rows,
    against the NumPy column builder of the drop-in package, bit for bit, written into records with a stride."""
    from muse_psfr_amd import _lib
    from muse_psfr_amd.psfrec import _fit_columns
    rng = np.random.default_rng(5)
    n = 37
    fit = rng.uniform(0.5, 3.0, (n, _lib.NFIT))
    fit[3, 4] = 1.0                               # n = 1: the flux error divides by n - 1
    fit[5, 0] = 0.0                               # peak = 0
    out = np.full((n, 20), -1.0)
    with np.errstate(all='ignore'):
        _lib.fit_rows(fit, 0.2, out[:, 1:15])
        cols = _fit_columns(np.zeros(n), fit, 0.2)
    want = np.column_stack([cols['center'], cols['flux'], cols['fwhm'], cols['n'], cols['peak'], cols['err_center'],
                            cols['err_flux'], cols['err_fwhm'], cols['err_n'], cols['err_peak']])
    assert want.shape == (n, 14)
    np.testing.assert_array_equal(out[:, 1:15], want)
    assert np.all(out[:, 0] == -1.0) and np.all(out[:, 15:] == -1.0)        # nothing outside the 14 columns
