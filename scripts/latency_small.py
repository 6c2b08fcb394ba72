"""Latency of the drop-in entry points on small inputs (what an interactive caller waits for):
compute_psf (one row) and compute_psf_from_sparta on tables of 1 .. 30 rows, native grid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muse_psfr_amd as M
from muse_psfr_amd import _minifits as mf
lb = np.linspace(490, 930, 35)
M.compute_psf(lb, 1.0, 0.7, 25., verbose=False)
ts = []
for _ in range(20):
    t = time.perf_counter(); M.compute_psf(lb, 1.0, 0.7, 25., verbose=False); ts.append(time.perf_counter() - t)
print('compute_psf, 35 lambda, 1280^2: median %.3f ms (min %.3f)' % (sorted(ts)[10] * 1e3, min(ts) * 1e3))
for n in (1, 4, 10, 30):
    see, gl, l0 = M.synthetic_rows(n)
    tbl = M.create_sparta_table(nlines=n)
    for k in range(1, 5):
        tbl.data['LGS%d_SEEING' % k][:] = see
        tbl.data['LGS%d_TUR_GND' % k][:] = gl
        tbl.data['LGS%d_L0' % k][:] = l0
    mk = lambda: mf.HDUList([mf.PrimaryHDU(), tbl])
    M.compute_psf_from_sparta(mk(), verbose=False)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); M.compute_psf_from_sparta(mk(), verbose=False); ts.append(time.perf_counter() - t)
    print('compute_psf_from_sparta, %2d rows x 35 lambda, 1280^2: median %.3f ms (min %.3f)' % (n, sorted(ts)[10] * 1e3, min(ts) * 1e3))
