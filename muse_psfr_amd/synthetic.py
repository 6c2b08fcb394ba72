"""Synthetic SPARTA inputs shared by bench.py, the tests and the golden generator.

The distribution is the one fixed in SURVEY.md section 8(d): per row seeing ~ U(0.4, 1.6) arcsec,
GL ~ U(0.30, 0.95), L0 ~ U(9, 29) m -- inside the validity window GL > 0, 8 < L0 < 30 of
the reference (psfrec.py:1049-1051) -- from ``np.random.default_rng(355)``.
"""
import numpy as np

SEED = 355


def synthetic_rows(nrows, seed=SEED):
    """Return (seeing, GL, L0) float64 arrays of length nrows (first nrows draws of the stream)."""
    u = np.random.default_rng(seed).random((int(nrows), 3))
    seeing = 0.4 + 1.2 * u[:, 0]
    gl = 0.30 + 0.65 * u[:, 1]
    l0 = 9.0 + 20.0 * u[:, 2]
    return seeing, gl, l0


def grid_pixscale(dim):
    """Pixel scale that makes an N x N grid runnable down to 465 nm (SURVEY.md 8(d) rule 1):
    0.2 arcsec * N / 1344, except the reference-native N = 1280 which keeps 0.2 arcsec."""
    return 0.2 if int(dim) == 1280 else 0.2 * int(dim) / 1344
