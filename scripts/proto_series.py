"""Prototype (NumPy) of the round-4 stage A: D_phi0 = r0^(-5/3) * poly(delta; Hd_k) + patch term.
Checks the Taylor series in delta = 1/L0^2 - eps0 of the fitting term and the 80x80 patch transform
against oracle.structure_function0(residual_psd)."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
import psfr_oracle as O
from numpy.fft import fft2, ifft2, fftshift

def binom(a, k):
    r = 1.0
    for i in range(k):
        r *= (a - i) / (i + 1)
    return r

def run(dim, seeing, GL, L0, K=7, eps0=1/128.):
    h = (100, 10000)
    tabs = O.ao_tables(h, False, 1, exact_masks=True)
    psd = O.residual_psd([GL, 1 - GL], h, seeing, L0, 1, dim, False, tables=tabs)[0]
    d_ref = O.structure_function0(psd)
    # --- series
    L = 2 * O.DPUP
    fx, fy = fftshift((np.mgrid[:dim, :dim] - (dim - 1) / 2) / L, axes=(1, 2))
    f2 = fx ** 2 + fy ** 2
    sel = f2 >= 1.5 ** 2
    cst = ((math.gamma(11 / 6) ** 2 / (2 * np.pi ** (11 / 3))) * (24 * math.gamma(6 / 5) / 5) ** (5 / 6))
    unit = (0.5 * 1000 / (2 * np.pi)) ** 2
    Hd = []
    for k in range(K):
        b = np.zeros_like(f2)
        b[sel] = cst * binom(-11 / 6, k) * (f2[sel] + eps0) ** (-11 / 6 - k)
        bg = fft2(b).real / L ** 2 * unit     # fft layout already (fx is fftshifted)
        Hd.append(2 * (bg[0, 0] - bg))
    r0 = O.seeing_to_r0(seeing)
    delta = 1 / L0 ** 2 - eps0
    dF = np.zeros_like(f2)
    for k in reversed(range(K)):
        dF = dF * delta + Hd[k]
    dF *= r0 ** (-5 / 3)
    # --- patch: P = psd - fit (fft layout), nonzero in the 80x80 zone only
    fit = np.zeros_like(f2)
    fit[sel] = cst * r0 ** (-5 / 3) * (f2[sel] + 1 / L0 ** 2) ** (-11 / 6) * unit
    P = fftshift(psd) - fit
    P[np.abs(P) < 1e-9 * np.abs(P).max()] = 0
    nz = np.argwhere(P != 0)
    su = np.where(nz[:, 0] < dim // 2, nz[:, 0], nz[:, 0] - dim)
    sv = np.where(nz[:, 1] < dim // 2, nz[:, 1], nz[:, 1] - dim)
    assert su.min() >= -40 and su.max() < 40 and sv.min() >= -40 and sv.max() < 40, (su.min(), su.max())
    bgP = fft2(P).real / L ** 2
    dP = 2 * (bgP[0, 0] - bgP)
    d_new = dF + dP
    err = np.abs(d_new - d_ref).max() / np.abs(d_ref).max()
    return err, np.abs(d_ref).max(), np.abs(dP).max(), np.abs(dF).max(), (P < 0).sum()

for dim in (256, 512):
    for (s, g, l0) in ((1.0, 0.7, 25.), (1.5, 0.3, 10.), (0.5, 0.9, 29.), (2.0, 0.5, 20.), (0.4, 0.5, 8.1), (1.0, 0.5, 1000.)):
        for K in (5, 6, 7, 8):
            print(dim, s, g, l0, K, run(dim, s, g, l0, K))
