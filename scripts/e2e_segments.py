"""Where compute_psf_from_sparta spends its wall time (tottime per function, several repeats)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muse_psfr_amd as M
from muse_psfr_amd import _minifits as mf
nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
see, gl, l0 = M.synthetic_rows(nrows)
tbl = M.create_sparta_table(nlines=nrows)
for k in range(1, 5):
    tbl.data['LGS%d_SEEING' % k][:] = see
    tbl.data['LGS%d_TUR_GND' % k][:] = gl
    tbl.data['LGS%d_L0' % k][:] = l0
mk = lambda: mf.HDUList([mf.PrimaryHDU(), tbl])
kw = dict(verbose=False) if dim == 1280 else dict(verbose=False, dim=dim, pixscale=M.grid_pixscale(dim), lmin=465, lmax=930)
for _ in range(3):
    M.compute_psf_from_sparta(mk(), **kw)
ts = []
for _ in range(10):
    t = time.perf_counter(); M.compute_psf_from_sparta(mk(), **kw); ts.append(time.perf_counter() - t)
print('%d rows dim %d: min %.2f ms median %.2f ms' % (nrows, dim, min(ts) * 1e3, sorted(ts)[5] * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    M.compute_psf_from_sparta(mk(), **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(16); print(s.getvalue()[:4000])
