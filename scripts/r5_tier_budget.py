"""CPU study: the tier masses of scripts/r5_tier_mass.py against a lower bound of the PSF peak that is cheap
to have before the masks are built: S_lb = the OTF summed exactly over the first NL lines (both half planes)."""
import sys
import numpy as np
sys.path.insert(0, 'oracle')
sys.path.insert(0, '.')
import psfr_oracle as O
from muse_psfr_amd.synthetic import grid_pixscale, synthetic_rows

H = (100, 10000)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nrow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tabs = O.ao_tables(H, False, 1, exact_masks=True)
see, gl, l0 = synthetic_rows(nrow)
extra = [(0.3, 0.98, 29.9), (0.6, 0.5, 15.0), (0.3, 0.5, 29.0), (2.5, 0.02, 29.9), (0.5, 0.9, 20.0)]
rows = list(zip(see, gl, l0)) + extra
lbs = np.linspace(465, 930, 35) if dim != 1280 else np.linspace(490, 930, 35)
tel = O.telescope_otf(dim)
tel = tel / tel[0, 0]
H1 = dim // 2 + 1
nmt = (H1 + 15) // 16
res = []
for (s, g, l) in rows:
    psd = O.residual_psd([g, 1 - g], H, s, l, 1, dim, False, tabs)[0]
    d0 = np.maximum(O.structure_function0(psd), 0)
    for lb in lbs:
        c = -0.5 * (2 * np.pi / lb) ** 2
        otf = tel * np.exp(c * d0)
        pad = np.zeros((nmt * 16, dim))
        pad[:H1] = otf[:H1]
        bmax = pad.reshape(nmt, 16, dim // 32, 32).max(axis=(1, 3))
        with np.errstate(divide='ignore'):
            e = np.log2(bmax)
        slb = otf[0].sum() + 2 * otf[1:NL].sum()
        fl = (e < -29.01) & (e > -49.0)
        mid = (e < -18.01) & (e >= -29.01)
        res.append((s, g, l, lb, otf.sum(), slb, 1024 * bmax[fl].sum(), 1024 * bmax[mid].sum() * 2.0 ** -11, fl.sum(), mid.sum()))
r = np.array(res)
tot, slb, mf, mm = r[:, 4], r[:, 5], r[:, 6], r[:, 7]
print('rows %d x %d wavelengths, dim %d, NL %d' % (len(rows), len(lbs), dim, NL))
print('peak/S_lb: median %.1f max %.1f' % (np.median(tot / slb), (tot / slb).max()))
for name, m in (('floor', mf), ('mid', mm), ('both', mf + mm)):
    print('%-5s mass/peak: max %.2e | mass/S_lb: max %.2e, 99%% %.2e, 90%% %.2e, median %.2e' % (
        name, (m / tot).max(), (m / slb).max(), np.percentile(m / slb, 99), np.percentile(m / slb, 90), np.median(m / slb)))
for eps in (2.5e-7, 5e-7, 1e-6, 2e-6, 5e-6):
    print('budget %.1e of S_lb: floor violated in %.1f %%, mid in %.1f %% of the (row, wavelength) pairs' % (
        eps, 100 * (mf > eps / 2 * slb).mean(), 100 * (mm > eps / 2 * slb).mean()))
i = np.argmax((mf + mm) / slb)
print('worst:', r[i])
