"""CPU oracle for the muse-psfr PSF-reconstruction hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker* for the HIP path in ``muse_psfr_amd``: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The
product never calls it (``muse_psfr_amd`` raises if ``libmpsfr.so`` is missing).

It is a NumPy restatement (fp64 throughout, same operation order where the order is observable)
of the reference's algorithm; every function cites the reference lines it follows
(``psfrec.py`` = /root/reference/muse_psfr/psfrec.py).  Parity pinning: ``oracle/make_golden.py``
ran the *real* reference in the build container and asserted agreement with this file to
<= 1e-12 (relative) on every stage before writing ``tests/golden/*.npz``; the GPU box re-checks
this file against those fixtures (``tests/test_oracle.py``).

The Moffat fit is *not* reference code: the reference calls ``mpdaf.obj.Image.moffat_fit``
(psfrec.py:863-865), an un-vendored dependency with no pinned version.  ``moffat_fit`` below is
the 5-parameter, unweighted, background-free circular least-squares definition (SURVEY.md
Appendix A) solved with MINPACK (scipy.optimize.leastsq); the reference's own tests pin it only
to +-1e-2 (test_psfrec.py:28-30, 121-127) -- those known answers are checked in tests/.

Two shapes are provided:
  * ``reference_shaped``  -- 4 complex FFTs per (direction, wavelength) exactly like
    psfrec.py:689-807; this is what bench.py times as the CPU baseline.
  * ``restructured``      -- the algebraically identical form the GPU uses (AO tables, one
    structure-function FFT per direction, OTF averaged over directions, pruned bilinear
    sampling).  Used to debug GPU intermediates; agrees with ``reference_shaped`` to ~1e-13.
"""
import math
import os

import numpy as np
from numpy.fft import fft2, fftshift, ifft2

MIN_L0 = 8    # psfrec.py:30
MAX_L0 = 30   # psfrec.py:31

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'muse_psfr_amd', 'data')

# system constants, psfrec.py:70-84, 103, 132-133
DPUP = 8.0
NACT = 24.0
ALT_DM = 1.0
FSAMP = 1000.0
DELAY_MS = 2.5
SEP_LGS = 63.0
NOISE_LGS = 1.0
DIM_AO = 80                      # 2 * Dimpup, psfrec.py:103, 138
WIND_DIR = (0.628163, -0.326497)  # psfrec.py:66
ARCMIN_H = 60 / 206265           # psfrec.py:279, 440


def seeing_to_r0(seeing, lbda_um=0.5, zenith=0.0):
    """psfrec.py:183-187."""
    r00p5 = 0.976 * 0.5 / seeing / 4.85
    return r00p5 * (lbda_um * 2) ** (6 / 5) * np.cos(np.deg2rad(zenith)) ** (3 / 5)


def lgs_positions(three_lgs_mode):
    """psfrec.py:86-93 -- (2, n_lgs) arcsec; 3-LGS mode always drops [1,-1]."""
    p = [[1, 1], [-1, -1], [-1, 1]] if three_lgs_mode else [[1, 1], [-1, -1], [-1, 1], [1, -1]]
    return np.array(p, dtype=float).T * SEP_LGS


def eval_directions(npsflin, field_size=60):
    """psfrec.py:154-158 -- (2, npsflin**2) arcsec."""
    x, y = (np.mgrid[:npsflin, :npsflin] - npsflin // 2) * field_size / 2
    return np.array([x, y]).reshape(2, -1)


def wind_speed_for(h):
    """psfrec.py:61 -- np.full_like(h, 12.5): 12 for integer altitudes, 12.5 for float ones."""
    return np.full_like(np.array(h), 12.5)


def _ao_freqs():
    """psfrec.py:548-554 and :241-242 -- folded frequency grid of the AO-corrected zone."""
    fx = np.fft.fftfreq(DIM_AO, DPUP / (DIM_AO // 2))[:, np.newaxis]
    fy = fx.T
    f = np.sqrt(fx ** 2 + fy ** 2)
    with np.errstate(all='ignore'):
        arg = fy / fx
    arg[0, 0] = 0
    arg = np.arctan(arg)
    return f, f * np.cos(arg), f * np.sin(arg)


def ao_tables(h, three_lgs_mode, npsflin, exact_masks=False, masks=None):
    """Row-independent part of dsp4muse (psfrec.py:531-613, :218-364, :367-528).

    Returns (T, noise): T[layer, dir, 80, 80] = |proj_layer|**2 and noise[dir, 80, 80] =
    err_noise, *before* the final transpose of psfrec.py:613, in FFT layout, so that
    ``dsp_res[dir] = sum_l T[l, dir] * Cphi_l + noise[dir]`` (psfrec.py:489, 515, 523).

    The cut-off masks of psfrec.py:257/:435 compare ``f*cos(arctan(fy/fx))`` with fc = 1.5 = 24/16
    exactly on the |fx| = 24/16 and |fy| = 24/16 lines, so in the reference their outcome there is
    decided by the last-bit rounding of libm/NumPy's cos/sin/arctan: it differs between NumPy
    1.26 and 2.2 on the same machine (44 vs 52 flipped pixels, measured).  ``exact_masks=False``
    reproduces whatever this interpreter's NumPy rounds to (bit-faithful to the reference run with
    the same NumPy); ``exact_masks=True`` applies the intended rule |k| >= 24 (reconstructor) /
    |k| > 24 (residual) on the integer grid -- the platform-independent definition the HIP path
    implements when it is given no masks.  ``masks=(rec, res)`` (two 80x80 boolean arrays in the
    reference's [i_fx][j_fy] indexing) imposes a captured outcome, e.g. the one of the NumPy that
    generated tests/golden (g1_ao_zone.npz).  The choice moves (fwhm, beta) by up to 3e-3
    (measured, DESIGN.md "cut-off masks"), far above the 1e-4 parity tolerance, hence the knob.
    """
    h = np.array(h)
    vent = wind_speed_for(h)
    f, f_x, f_y = _ao_freqs()
    poslgs = lgs_positions(three_lgs_mode) / 60          # arcmin, psfrec.py:536
    dirperf = eval_directions(npsflin) / 60
    ngs = poslgs.shape[1]
    pitch = DPUP / NACT
    fc = 1 / (2 * pitch)
    sig = NOISE_LGS

    def shack(strict):
        # psfrec.py:252-257 (>=, reconstructor) and :430-435 (>, residual); note the missing
        # parentheses: ((f != 0) & (|fx| ? fc)) | (|fy| ? fc)
        w = 2 * np.pi * 1j * f * np.sinc(pitch * f_x) * np.sinc(pitch * f_y)
        if masks is not None:
            m = np.asarray(masks[1 if strict else 0], dtype=bool).reshape(DIM_AO, DIM_AO)
        elif exact_masks:
            # platform-independent rule on the *unfolded* integer frequency grid (see docstring)
            k = np.abs(np.fft.fftfreq(DIM_AO, 1 / DIM_AO).astype(int))
            kc = int(round(fc * 2 * DPUP))
            m = ((k[:, None] > kc) | (k[None, :] > kc)) if strict else \
                ((k[:, None] >= kc) | (k[None, :] >= kc))
        elif strict:
            m = (f != 0) & (np.abs(f_x) > fc) | (np.abs(f_y) > fc)
        else:
            m = (f != 0) & (np.abs(f_x) >= fc) | (np.abs(f_y) >= fc)
        w[m] = 0.
        return w

    # --- reconstructor, psfrec.py:272-364 with LSE=True, one reconstructed layer at ALT_DM
    wfs = shack(False)
    Mr = np.zeros((ngs, DIM_AO, DIM_AO), dtype=complex)
    for j in range(ngs):
        ff_x = f_x * poslgs[0, j] * ALT_DM * ARCMIN_H
        ff_y = f_y * poslgs[1, j] * ALT_DM * ARCMIN_H
        Mr[j] = wfs * np.exp(1j * 2 * np.pi * (ff_x + ff_y))
    res_tmp = Mr.conj() * (1 / sig)
    MAP = np.sum(res_tmp * Mr, axis=0)
    with np.errstate(all='ignore'):
        inv = np.where(MAP != 0, 1 / MAP, 0)      # 1x1 inverse, psfrec.py:339-354
    inv[0, 0] = 0
    W = inv * res_tmp                             # (ngs, 80, 80)

    # --- residual, psfrec.py:400-523 with tempo=True
    wfs = shack(True)
    wind = np.stack([vent * np.cos(WIND_DIR), vent * np.sin(WIND_DIR)])
    ti = 1 / FSAMP
    td = DELAY_MS * 1e-3
    deltaT = ti + td
    nl = h.size
    ndir = dirperf.shape[1]
    T = np.zeros((nl, ndir, DIM_AO, DIM_AO))
    noise = np.zeros((ndir, DIM_AO, DIM_AO))
    Mv = np.zeros((nl, ngs, DIM_AO, DIM_AO), dtype=complex)
    for i in range(nl):
        for j in range(ngs):
            ff_x = f_x * poslgs[0, j] * h[i] * ARCMIN_H
            ff_y = f_y * poslgs[1, j] * h[i] * ARCMIN_H
            www = np.sinc(wind[0, i] * ti * f_x + wind[1, i] * ti * f_y)
            Mv[i, j] = www * wfs * np.exp(1j * 2 * (ff_x + ff_y) * np.pi)
    for d in range(ndir):
        beta = dirperf[:, d]
        pdm = np.exp(1j * 2 * np.pi * ALT_DM * ARCMIN_H * (beta[0] * f_x + beta[1] * f_y))
        ptmp = pdm * W                              # (ngs, 80, 80)
        for i in range(nl):
            pbeta = np.exp(1j * 2 * np.pi * (h[i] * ARCMIN_H * (beta[0] * f_x + beta[1] * f_y)
                                             - (wind[0, i] * deltaT * f_x + wind[1, i] * deltaT * f_y)))
            proj = pbeta - np.sum(ptmp * Mv[i], axis=0)
            T[i, d] = (proj * proj.conj()).real
        noise[d] = np.sum(ptmp * sig * ptmp.conj(), axis=0).real
        noise[d, 0, 0] = 0
    return T, noise


def numpy_cutoff_masks():
    """(rec, res) masks exactly as this interpreter's NumPy evaluates psfrec.py:257 and :435."""
    f, f_x, f_y = _ao_freqs()
    fc = 1 / (2 * DPUP / NACT)
    rec = (f != 0) & (np.abs(f_x) >= fc) | (np.abs(f_y) >= fc)
    res = (f != 0) & (np.abs(f_x) > fc) | (np.abs(f_y) > fc)
    return rec, res


def ao_zone_psd(Cn2, h, L0, r0, three_lgs_mode, npsflin, tables=None):
    """dsp4muse, psfrec.py:531-613: (ndir, 80, 80) residual PSD of the corrected zone."""
    Cn2 = np.atleast_1d(np.asarray(Cn2, dtype=float))
    f, _, _ = _ao_freqs()
    T, noise = tables if tables is not None else ao_tables(h, three_lgs_mode, npsflin)
    with np.errstate(all='ignore'):
        cphi = (0.0229 * (Cn2[:, None, None] ** (-3 / 5) * r0) ** (-5 / 3) *
                (f ** 2 + (1 / L0) ** 2) ** (-11 / 6))            # psfrec.py:569-571
    dsp = np.sum(T * cphi[:, None], axis=0)
    dsp[:, 0, 0] = 0                                              # psfrec.py:490
    dsp = dsp + noise
    return np.moveaxis(dsp, -1, -2)                               # psfrec.py:613


def fitting_psd(dim, L, r0, L0, fc):
    """psd_fit, psfrec.py:616-626 (half-pixel-offset grid, FFT layout)."""
    dim = int(dim)
    fx, fy = fftshift((np.mgrid[:dim, :dim] - (dim - 1) / 2) / L, axes=(1, 2))
    f = np.sqrt(fx ** 2 + fy ** 2)
    out = np.zeros_like(f)
    cst = ((math.gamma(11 / 6) ** 2 / (2 * np.pi ** (11 / 3))) *
           (24 * math.gamma(6 / 5) / 5) ** (5 / 6))
    sel = f >= fc
    out[sel] = cst * r0 ** (-5 / 3) * (f[sel] ** 2 + (1 / L0) ** 2) ** (-11 / 6)
    return out


def residual_psd(Cn2, h, seeing, L0, npsflin=1, dim=1280, three_lgs_mode=False, tables=None):
    """simul_psd_wfm, psfrec.py:36-151: (ndir, dim, dim) PSD [nm^2 m^2], DC at [dim/2, dim/2]."""
    Cn2 = np.array(Cn2, dtype=float)
    Cn2 /= Cn2.sum()
    r0 = seeing_to_r0(seeing, 0.5, 0.0)
    fc = 1 / (2 * DPUP / NACT)
    dsp = ao_zone_psd(Cn2, h, L0, r0, three_lgs_mode, npsflin, tables)
    dspa = fftshift(fitting_psd(dim, 2 * DPUP, r0, L0, fc))
    dspf = np.resize(dspa, (dsp.shape[0], dim, dim))
    half = DIM_AO // 2
    sl = slice(dim // 2 - half, dim // 2 + half)
    dspf[:, sl, sl] = np.maximum(dspa[sl, sl], fftshift(dsp, axes=(1, 2)))
    return dspf * (0.5 * 1000 / (2 * np.pi)) ** 2


def pupil_mask(radius, width, oc=0.0):
    """psfrec.py:190-203."""
    c = (width - 1) / 2
    x, y = np.ogrid[:width, :width]
    rho = np.hypot(x - c, y - c) / radius
    return ((rho < 1) & (rho >= oc)).astype(int)


def npix_crop(lbda_nm, dimpsf=40, pixscale=0.2):
    """psfrec.py:663-664."""
    return (np.round(((dimpsf * pixscale * 2 * 8 * 4.85 * 1000) / lbda_nm) / 2) * 2).astype(int)


def telescope_otf(dim):
    """psfrec.py:784-790, FFT layout (DC at [0,0]); wavelength- and row-independent."""
    pup = pupil_mask(dim / 4, dim // 2, oc=0.14)
    tab = np.zeros((dim, dim), dtype=complex)
    tab[:dim // 2, :dim // 2] = pup
    return np.abs(fft2(np.abs(ifft2(tab)) ** 2)) / pup.sum()


def psd_to_psf_refshaped(psd, pup, lbda_m):
    """psd_to_psf, psfrec.py:689-807 with samp=2=sampnum and FoV=FoVnum (the only live branch)."""
    dim = psd.shape[0]
    npup = pup.shape[0]
    L = DPUP * (dim / npup)
    convnm = 2 * np.pi / (lbda_m * 1e9)
    bg = ifft2(fftshift(psd * convnm ** 2)) * (psd.size / L ** 2)     # FFT 1
    Dphi = fftshift(2 * (bg[0, 0].real - bg.real))
    tab = np.zeros((dim, dim), dtype=complex)
    tab[:npup, :npup] = pup
    dlFTO = fft2(np.abs(ifft2(tab)) ** 2)                             # FFT 2, 3
    dlFTO = fftshift(np.abs(dlFTO) / pup.sum())
    sysFTO = fftshift(np.exp(-0.5 * Dphi) * dlFTO)
    sysPSF = np.real(fftshift(ifft2(sysFTO)))                         # FFT 4
    return sysPSF / sysPSF.sum()


def _bilinear_sample(psf, dimpsf):
    """interpolate(), psfrec.py:635-641 + :682-683: out[i,j] = bilinear(psf, (i*s, j*s))."""
    n = psf.shape[0]
    pos = np.arange(dimpsf) * n / dimpsf
    i0 = np.minimum(np.floor(pos).astype(int), n - 2)
    w = pos - i0
    a = psf[i0][:, i0]
    b = psf[i0 + 1][:, i0]
    c = psf[i0][:, i0 + 1]
    d = psf[i0 + 1][:, i0 + 1]
    wi = w[:, None]
    wj = w[None, :]
    return (a * (1 - wi) * (1 - wj) + b * wi * (1 - wj) + c * (1 - wi) * wj + d * wi * wj)


def psf_stamps_refshaped(psd, lbda_nm, dimpsf=40, pixscale=0.2):
    """psf_muse, psfrec.py:644-686.  psd: (dim,dim) or (ndir,dim,dim) -> (nl, dimpsf, dimpsf)."""
    lbda_nm = np.atleast_1d(np.asarray(lbda_nm, dtype=float))
    if psd.ndim == 2:
        psd = psd[None]
    ndir, dim = psd.shape[0], psd.shape[1]
    pup = pupil_mask(dim / 4, dim // 2, oc=0.14)
    npixc = npix_crop(lbda_nm, dimpsf, pixscale)
    if npixc.max() > dim:
        raise ValueError('grid too small: npixc=%d > dim=%d' % (npixc.max(), dim))
    out = np.zeros((lbda_nm.size, dimpsf, dimpsf))
    for i, lb in enumerate(lbda_nm):
        half = npixc[i] // 2
        sl = slice(dim // 2 - half, dim // 2 + half)
        psf = np.zeros((npixc[i], npixc[i]))
        for j in range(ndir):
            psf += psd_to_psf_refshaped(psd[j], pup, lb * 1e-9)[sl, sl]
        psf /= ndir
        psf /= psf.sum()
        np.maximum(psf, 0, out=psf)
        out[i] = _bilinear_sample(psf, dimpsf)
    out /= out.sum(axis=(1, 2))[:, None, None]
    return out


def muse_intrinsic_psf(lbda_nm):
    """psfrec.py:1144-1171 (fwhm arcsec, beta)."""
    pol_beta = [-0.83704697, 1.1337153, 0.0609222, -1.35581762, 1.15237178, 2.2106042]
    pol_fwhm = [0.60467385, -1.58905792, 1.75293264, -1.0368302, 0.21487023, 0.34851139]
    lb = (10 * np.asarray(lbda_nm, dtype=float) - 4750) / (9350 - 4750)
    return np.polyval(pol_fwhm, lb), np.polyval(pol_beta, lb)


def moffat_kernel(gamma, alpha, size):
    """astropy.convolution.Moffat2DKernel(gamma, alpha, x_size=size, y_size=size) (astropy 4.3.1
    convolution/kernels.py:814-821): model sampled at integer offsets, normalised to sum 1."""
    r = np.arange(size) - size // 2
    rr = r[:, None] ** 2 + r[None, :] ** 2
    k = (1 + rr / gamma ** 2) ** (-alpha)
    return k / k.sum()


def convolve_same(img, ker):
    """scipy.signal.fftconvolve(img, ker, mode='same') restated as the direct zero-padded linear
    convolution cropped to the centre (identical up to FFT rounding ~1e-17)."""
    n = img.shape[0]
    k = ker.shape[0]
    c = k // 2
    out = np.zeros_like(img)
    for di in range(k):
        for dj in range(k):
            w = ker[di, dj]
            # out[i,j] += w * img[i - (di-c), j - (dj-c)]
            si, sj = di - c, dj - c
            i0, i1 = max(0, si), min(n, n + si)
            j0, j1 = max(0, sj), min(n, n + sj)
            if i0 < i1 and j0 < j1:
                out[i0:i1, j0:j1] += w * img[i0 - si:i1 - si, j0 - sj:j1 - sj]
    return out


def load_coeff_l0():
    """coeffL0.fits (psfrec.py:895-896) converted to .npy: (2, 200) float32 [L0 grid, coeff]."""
    return np.load(os.path.join(_DATA, 'coeffL0.npy'))


def tiptilt_alpha(seeing, GL, L0, pixscale=0.2, coeff=None):
    """psfrec.py:879-905: Moffat(beta=2) alpha [pixels] of the residual tip-tilt kernel."""
    if coeff is None:
        coeff = load_coeff_l0()
    seeingHL = seeing * (1 - GL) ** (3. / 5.)
    r0HL = 0.976 * 0.5 / seeingHL / 4.85
    coeffHL = np.interp(L0, coeff[0], coeff[1])
    fwhmTT = (np.sqrt(coeffHL * 0.97 * 6.88 * (.5 * 1.e-6 / (2. * np.pi)) ** 2 *
                      8 ** (-1 / 3.) * r0HL ** (-5 / 3.)) / (4.85 * 1.e-6) * 2.35 / pixscale)
    return fwhmTT / (2 * np.sqrt(2 ** (1. / 2) - 1))


def convolve_final_psf(lbda_nm, seeing, GL, L0, psf, pixscale=0.2, coeff=None):
    """psfrec.py:874-930."""
    lbda_nm = np.atleast_1d(np.asarray(lbda_nm, dtype=float))
    n = psf.shape[1]
    ks = n + 1 if n % 2 == 0 else n
    ktt = moffat_kernel(tiptilt_alpha(seeing, GL, L0, pixscale, coeff), 2, ks)
    fwhm, beta = muse_intrinsic_psf(lbda_nm)
    fwhm = fwhm / pixscale
    alpha = fwhm / (2 * np.sqrt(2 ** (1. / beta) - 1))
    out = np.zeros_like(psf)
    for k in range(lbda_nm.size):
        tmp = convolve_same(psf[k], ktt)
        out[k] = convolve_same(tmp, moffat_kernel(alpha[k], beta[k], ks))
    return out


def moffat_model(v, P, Q):
    return v[0] * (1 + ((P - v[1]) / v[3]) ** 2 + ((Q - v[2]) / v[3]) ** 2) ** (-v[4])


def moffat_fit(im, pixscale=0.2, full=False, errors=False):
    """Stand-in for mpdaf Image.moffat_fit(circular=True, fit_back=False) (psfrec.py:863-865),
    SURVEY.md Appendix A.  Returns (peak, p0, q0, fwhm_arcsec, beta).

    errors=True returns a dict with, besides those five, the columns psfrec.py:866-870 keeps from
    mpdaf's result object: flux, err_peak, err_center, err_n, err_fwhm (arcsec), err_flux (and
    alpha, err_alpha, chi2, dof).  mpdaf's published error recipe: MINPACK's `cov_x` (the inverse
    of J^T J at the solution) scaled by the reduced chi square,
        err_i = sqrt(|cov_x[i, i]| * chi2 / dof),    dof = npix - 5,
    for the fitted variables (I, p0, q0, a, n); flux = I pi a^2 / (n - 1) (the integral of the
    circular Moffat).  Derived quantities by first-order propagation: err_fwhm through
    fwhm = 2 a sqrt(2^(1/n) - 1) with the (a, n) block of the covariance (the two are strongly
    anti-correlated: the variance of the FWHM is what a fit in (fwhm, n) would report directly);
    err_flux from the relative errors of I, a^2 and (n - 1) added in quadrature.  mpdaf is absent
    (SURVEY.md 8c): the formulas mpdaf itself uses for those two derived columns are NOT pinned
    by anything in the reference -- "parity unpinned" for err_fwhm / err_flux beyond this
    definition."""
    from scipy.optimize import leastsq
    P, Q = np.indices(im.shape)
    P = P.ravel().astype(float)
    Q = Q.ravel().astype(float)
    d = im.ravel()
    c = np.unravel_index(im.argmax(), im.shape)
    n0 = 2.0
    a0 = 4.0 / (2 * np.sqrt(2 ** (1 / n0) - 1))
    v, cov, info, _, ier = leastsq(lambda v: moffat_model(v, P, Q) - d,
                                   [im[c], c[0], c[1], a0, n0], full_output=True,
                                   xtol=1e-14, ftol=1e-14, gtol=0.0)
    a, n = abs(v[3]), v[4]
    s2 = 2 ** (1 / n) - 1
    fwhm = 2 * a * np.sqrt(s2) * pixscale
    res = (v[0], v[1], v[2], fwhm, n)
    chi2 = float(np.sum(info['fvec'] ** 2))
    if errors:
        dof = d.size - v.size
        s = abs(chi2 / dof)
        out = dict(peak=v[0], center=np.array([v[1], v[2]]), fwhm=fwhm, n=n, alpha=a, chi2=chi2, dof=dof,
                   flux=v[0] * np.pi * a * a / (n - 1))
        if cov is None:                 # singular J^T J (flat valley): mpdaf then reports no errors
            out.update(err_peak=np.nan, err_center=np.full(2, np.nan), err_alpha=np.nan, err_n=np.nan,
                       err_fwhm=np.nan, err_flux=np.nan)
            return out
        err = np.sqrt(np.abs(np.diag(cov)) * s)
        # d fwhm / d(a, n), pixels
        g = np.array([2 * np.sqrt(s2) * np.sign(v[3]),
                      -a * 2 ** (1 / n) * np.log(2) / (n * n * np.sqrt(s2))])
        var_fw = g @ cov[3:5, 3:5] @ g
        with np.errstate(all='ignore'):
            relf = np.sqrt((err[0] / v[0]) ** 2 + (2 * err[3] / a) ** 2 + (err[4] / (n - 1)) ** 2)
        out.update(err_peak=err[0], err_center=err[1:3].copy(), err_alpha=err[3], err_n=err[4],
                   err_fwhm=np.sqrt(abs(var_fw) * s) * pixscale, err_flux=abs(out['flux']) * relf)
        return out
    if full:
        return res, chi2, info['nfev']
    return res


def fit_psf_cube(psf, pixscale=0.2):
    """psfrec.py:861-871 -> (nl, 5) array [peak, p0, q0, fwhm", beta]."""
    return np.array([moffat_fit(p, pixscale) for p in psf])


def compute_psf(lbda_nm, seeing, GL, L0, npsflin=1, h=(100, 10000), three_lgs_mode=False,
                dim=1280, dimpsf=40, pixscale=0.2, tables=None, coeff=None, fit=True):
    """compute_psf, psfrec.py:933-978, reference-shaped.  Returns (fit[nl,5] | None, psf)."""
    lbda_nm = np.atleast_1d(np.asarray(lbda_nm, dtype=float))
    psd = residual_psd([GL, 1 - GL], h, seeing, L0, npsflin, dim, three_lgs_mode, tables)
    psf = psf_stamps_refshaped(psd, lbda_nm, dimpsf, pixscale)
    psf = convolve_final_psf(lbda_nm, seeing, GL, L0, psf, pixscale, coeff)
    return (fit_psf_cube(psf, pixscale) if fit else None), psf


# ----------------------------------------------------------------------------------------------
# restructured form (what the GPU computes); SURVEY.md Appendix B identities
# ----------------------------------------------------------------------------------------------

def structure_function0(psd):
    """D_phi0 in FFT layout with D_phi(lambda) = (2 pi / lambda_nm)^2 * D_phi0
    (psfrec.py:717-722 with the lambda factor pulled out).  psd: centred (dim, dim)."""
    dim = psd.shape[0]
    L = 2 * DPUP
    bg = ifft2(fftshift(psd)).real * (psd.size / L ** 2)
    return 2 * (bg[0, 0] - bg)


def sample_matrix(dim, npixc, dimpsf):
    """G[i, u] = (1-a_i) W^(u x_i) + a_i W^(u (x_i+1)), W = exp(2 pi i / dim): the bilinear
    sampling of psfrec.py:682-683 folded into the inverse DFT of psfrec.py:800 (x_i = native-layout
    index of the left neighbour of sample i in the centred crop of psfrec.py:672-677)."""
    q = np.arange(dimpsf) * npixc
    i0 = q // dimpsf
    a = (q % dimpsf) / dimpsf
    x0 = (i0 - npixc // 2) % dim
    u = np.arange(dim)
    w0 = np.exp(2j * np.pi * ((u[None, :] * x0[:, None]) % dim) / dim)
    w1 = np.exp(2j * np.pi * ((u[None, :] * (x0[:, None] + 1)) % dim) / dim)
    return (1 - a)[:, None] * w0 + a[:, None] * w1


def psf_stamps_restructured(psd, lbda_nm, dimpsf=40, pixscale=0.2):
    """Same result as psf_stamps_refshaped, via one D_phi0 per direction, the hoisted telescope
    OTF, the direction-averaged OTF and the pruned (sampled) inverse transform."""
    lbda_nm = np.atleast_1d(np.asarray(lbda_nm, dtype=float))
    if psd.ndim == 2:
        psd = psd[None]
    ndir, dim = psd.shape[0], psd.shape[1]
    tel = telescope_otf(dim)
    d0 = np.array([structure_function0(p) for p in psd])
    npixc = npix_crop(lbda_nm, dimpsf, pixscale)
    if npixc.max() > dim:
        raise ValueError('grid too small: npixc=%d > dim=%d' % (npixc.max(), dim))
    out = np.zeros((lbda_nm.size, dimpsf, dimpsf))
    for k, lb in enumerate(lbda_nm):
        s = (2 * np.pi / lb) ** 2
        otf = tel * np.exp(-0.5 * s * d0).sum(axis=0)
        G = sample_matrix(dim, npixc[k], dimpsf)
        st = (G @ otf @ G.T).real
        st = np.maximum(st, 0)
        out[k] = st / st.sum()
    return out


def _split16(x, flush=True):
    """x = hi + lo in fp16 (round to nearest even), as the matrix-core kernel splits its operands;
    flush: fp16 subnormals set to zero -- the model the kernel was designed against; the gfx950 matrix
    cores themselves keep subnormal inputs (tests/test_gpu_parity.py, precision tiers), which only makes
    the product more accurate than this model."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    if flush:
        tiny = np.float16(2.0 ** -14)
        hi = np.where(np.abs(hi) < tiny, np.float16(0), hi)
        lo = np.where(np.abs(lo) < tiny, np.float16(0), lo)
    return hi.astype(np.float32), lo.astype(np.float32)


def psf_stamps_contraction_fp16(psd, lbda_nm, dimpsf=40, pixscale=0.2, otf_shift=15, tab_shift=9):
    """Arithmetic model of the matrix-core per-wavelength kernel (muse_psfr_amd/csrc/otf_mfma.hip):
    the stamps of psf_stamps_refshaped via the half plane v in [0, N/2], the 21 distinct samples
    per direction (E_(40-i) = conj E_i, so stamp = P +- Q), fp32 OTF elements 2^(c D + log2 tel),
    every operand split into two fp16 halves (three products, fp32 accumulation) and scaled by
    2^otf_shift / 2^tab_shift out of the fp16 subnormal range.  Test infrastructure: documents what
    precision the split buys (and what an unscaled table costs); the kernel itself is checked
    against psf_stamps_refshaped on the GPU."""
    lbda_nm = np.atleast_1d(np.asarray(lbda_nm, dtype=float))
    if psd.ndim == 2:
        psd = psd[None]
    dim = psd.shape[1]
    nh, ns = dim // 2 + 1, dimpsf // 2 + 1
    tel = telescope_otf(dim)
    tel = (tel / tel[0, 0])[:, :nh].T                        # tel[0][0] = 1; transposed half plane [v][u]
    with np.errstate(divide='ignore'):
        tl2 = (np.log2(tel) + otf_shift).astype(np.float32)
    d0 = np.array([structure_function0(p)[:, :nh].T for p in psd]).astype(np.float32)
    npixc = npix_crop(lbda_nm, dimpsf, pixscale)
    wv = np.full(nh, 2.0)
    wv[0] = wv[-1] = 1.0
    f32 = np.float32
    lg = int(np.ceil(np.log2(dim)))
    out = np.zeros((lbda_nm.size, dimpsf, dimpsf))
    for k, lb in enumerate(lbda_nm):
        c2 = f32(-0.5 * (2 * np.pi / lb) ** 2 * np.log2(np.e))
        a = np.exp2(c2 * d0 + tl2[None], dtype=f32).sum(axis=0, dtype=f32)       # [v][u], x 2^otf_shift
        a = a * f32(2.0 ** -np.ceil(np.log2(psd.shape[0])))
        S = sample_matrix(dim, npixc[k], dimpsf)[:ns]          # [i][x], W = exp(+2 pi i / N)
        E = np.conj(S).T * 2.0 ** tab_shift                    # forward kernel along the line: [u][i]
        G = (S[:, :nh] * wv[None, :]).T * 2.0 ** tab_shift     # conj of the forward kernel: [v][j]
        ah, al = _split16(a)
        t = []
        for part in (E.real, E.imag):                          # first contraction: Tq = OTF . E
            eh, el = _split16(part)
            t.append((al @ eh + ah @ el + ah @ eh).astype(f32))
        scale = f32(2.0 ** -(lg + tab_shift))
        pq = []
        for tq, part in zip(t, (G.real, G.imag)):              # second: P = Tx^T Gx, Q = Ty^T Gy
            th, tw = _split16(tq * scale)
            gh, gl = _split16(part)
            pq.append((tw.T @ gh + th.T @ gl + th.T @ gh).astype(f32))
        P, Q = pq
        st = np.zeros((dimpsf, dimpsf), f32)
        ii = np.arange(ns)
        st[np.ix_(ii, ii)] = P + Q
        m = np.arange(1, dimpsf // 2)                           # mirrored indices 40 - i
        st[np.ix_(dimpsf - m, ii)] = (P - Q)[m]
        st[np.ix_(ii, dimpsf - m)] = (P - Q)[:, m]
        st[np.ix_(dimpsf - m, dimpsf - m)] = (P + Q)[np.ix_(m, m)]
        st = np.maximum(st, 0)
        out[k] = st / st.sum(dtype=np.float64)
    return out
