"""CPU study (oracle, NumPy): the OTF mass that the precision tiers of the matrix-core stage leave out.
Per (row, wavelength): blocks of 16 lines x 32 columns of the OTF half plane, bound = max element of the block;
floor: blocks with bound < 2^-29.01 are dropped; mid: blocks with bound < 2^-18.01 lose the low fp16 half
(relative 2^-11 per element).  Prints the dropped mass relative to sum(OTF) = the PSF peak."""
import sys
import numpy as np
sys.path.insert(0, 'oracle')
sys.path.insert(0, '.')
import psfr_oracle as O
from muse_psfr_amd.synthetic import grid_pixscale

H = (100, 10000)


def study(dim, see, gl, l0, lbs, tabs):
    psd = O.residual_psd([gl, 1 - gl], H, see, l0, 1, dim, False, tabs)[0]
    d0 = O.structure_function0(psd)
    tel = O.telescope_otf(dim)
    tel = tel / tel[0, 0]                 # OTF[0][0] = 1: the unit of the thresholds
    H1 = dim // 2 + 1
    for lb in lbs:
        c = -0.5 * (2 * np.pi / lb) ** 2
        otf = tel * np.exp(c * np.maximum(d0, 0))
        half = otf[:H1]                       # lines v = 0 .. N/2 (weight 2 except the two self-conjugate lines)
        nmt = (H1 + 15) // 16
        pad = np.zeros((nmt * 16, dim))
        pad[:H1] = half
        blk = pad.reshape(nmt, 16, dim // 32, 32)
        bmax = blk.max(axis=(1, 3))
        bsum = blk.sum(axis=(1, 3))
        tot = otf.sum()
        with np.errstate(divide='ignore'):
            e = np.log2(bmax)
        fl = (e < -29.01) & (e > -49.0)
        mid = (e < -18.01) & (e >= -29.01)
        m_floor_true = 2 * bsum[fl].sum()
        m_floor_ub = 1024 * bmax[fl].sum()
        m_mid_ub = 1024 * bmax[mid].sum() * 2.0 ** -11
        print('dim %4d see %.2f gl %.2f l0 %4.1f lb %3.0f | sumOTF %8.1f | floor: %4d blocks, true %.2e ub %.2e of OTF00, '
              'ub/peak %.2e | mid: %4d blocks, ub %.2e, ub/peak %.2e | kept %.3f' % (
                  dim, see, gl, l0, lb, tot, fl.sum(), m_floor_true, m_floor_ub, m_floor_ub / tot, mid.sum(), m_mid_ub,
                  m_mid_ub / tot, (e >= -29.01).mean()))


if __name__ == '__main__':
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ps = grid_pixscale(dim) if dim != 1280 else 0.2
    tabs = O.ao_tables(H, False, 1, exact_masks=True)
    rows = [(1.0, 0.7, 25.0), (0.3, 0.98, 29.9), (0.3, 0.98, 8.1), (0.4, 0.95, 29.0), (0.5, 0.9, 20.0), (1.6, 0.3, 9.0), (2.5, 0.02, 29.9),
            (0.6, 0.5, 15.0), (0.3, 0.5, 29.0)]
    lbs = [465.0, 700.0, 930.0] if dim != 1280 else [490.0, 700.0, 930.0]
    for r in rows:
        study(dim, *r, lbs, tabs)
