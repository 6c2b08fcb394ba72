"""End-to-end time of the drop-in entry point on a synthetic SPARTA table: what a caller of
compute_psf_from_sparta waits for (FITS in memory -> HDUList out), and where it goes."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muse_psfr_amd as M
from muse_psfr_amd import _minifits as mf
from muse_psfr_amd.psfrec import _astropy

nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
see, gl, l0 = M.synthetic_rows(nrows)
tbl = M.create_sparta_table(nlines=nrows)
for k in range(1, 5):
    tbl.data['LGS%d_SEEING' % k][:] = see
    tbl.data['LGS%d_TUR_GND' % k][:] = gl
    tbl.data['LGS%d_L0' % k][:] = l0
fits, _ = _astropy()
mk = (lambda: fits.HDUList([fits.PrimaryHDU(), tbl])) if fits is not None else (lambda: mf.HDUList([mf.PrimaryHDU(), tbl]))
kw = dict(verbose=False) if dim == 1280 else dict(verbose=False, dim=dim, pixscale=M.grid_pixscale(dim), lmin=465, lmax=930)
M.compute_psf_from_sparta(mk(), **kw)            # warm: context, tables
t = time.perf_counter(); res = M.compute_psf_from_sparta(mk(), **kw); dt = time.perf_counter() - t
print('%d rows x 35 lambda, dim %d: %.1f ms per call = %.2f M PSFs/s end to end' % (nrows, dim, dt * 1e3, nrows * 35 / dt / 1e6))
pr = cProfile.Profile(); pr.enable(); M.compute_psf_from_sparta(mk(), **kw); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(s.getvalue()[:3500])
