"""Where the 1000-row call of compute_psf_from_sparta spends its wall time: the phases of
psfrec._reconstruct_pipelined with the wall clock, and the library's own host time per asynchronous call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muse_psfr_amd as M
from muse_psfr_amd import psfrec as P

n, dim = 1000, 512
ps = M.grid_pixscale(dim)
see, gl, l0 = M.synthetic_rows(n)
lb = np.linspace(465, 930, 35)
stats = np.stack([see, gl, l0], axis=1)
three = np.zeros(n, bool)
las = np.full(n, -1)
ctx = P.get_context(dim, ps, 40, 'mixed', 0, 0)
masks = P._resolve_masks('exact')
for nparts in (4, 2, 3, 5, 8):
    bounds = [n * k // nparts for k in range(nparts + 1)]
    t3 = three.astype(np.uint8)
    rows = []
    for rep in range(12):
        ctx.profile_reset()
        T = [time.perf_counter()]
        pend = [ctx.reconstruct_async(lb, see[a:b], gl[a:b], l0[a:b], t3[a:b], (100, 10000), npsflin=1, masks=masks, want_psf=False)
                for a, b in zip(bounds[:-1], bounds[1:])]
        T.append(time.perf_counter())
        rec, blk = P._fit_rows_template(lb, stats, las)
        T.append(time.perf_counter())
        tw = tf = 0.0
        for (a, b), p in zip(zip(bounds[:-1], bounds[1:]), pend):
            t0 = time.perf_counter(); r = p.wait(); t1 = time.perf_counter()
            P._fit_rows_fill(blk[a * 35:b * 35], r['fit'], ps); t2 = time.perf_counter()
            tw += t1 - t0; tf += t2 - t1
        T.append(time.perf_counter())
        hs, hc = ctx.host_time()
        rows.append((T[1] - T[0], T[2] - T[1], tw, tf, T[3] - T[0], hs / max(hc, 1)))
    r = np.median(np.array(rows[2:]), axis=0) * 1e3
    print('parts %d: queue %.3f  template %.3f  waits %.3f  fills %.3f  total %.3f ms | library host time per call %.3f ms' % (nparts, *r))
# one blocking call for comparison
ts = []
for _ in range(8):
    t0 = time.perf_counter(); ctx.reconstruct(lb, see, gl, l0, t3, (100, 10000), masks=masks, want_psf=False); ts.append(time.perf_counter() - t0)
print('one blocking call of 1000 rows: %.3f ms' % (np.median(ts) * 1e3))

# the whole entry point, top-level segments by cumulative time
import cProfile, io, pstats
from muse_psfr_amd import _minifits as mf
tbl = M.create_sparta_table(nlines=n)
for k in range(1, 5):
    tbl.data['LGS%d_SEEING' % k][:] = see
    tbl.data['LGS%d_TUR_GND' % k][:] = gl
    tbl.data['LGS%d_L0' % k][:] = l0
mk = lambda: mf.HDUList([mf.PrimaryHDU(), tbl])
kw = dict(verbose=False, dim=dim, pixscale=ps, lmin=465, lmax=930, cutoff_masks='exact')
for _ in range(3):
    M.compute_psf_from_sparta(mk(), **kw)
ts = []
for _ in range(20):
    t = time.perf_counter(); M.compute_psf_from_sparta(mk(), **kw); ts.append(time.perf_counter() - t)
print('compute_psf_from_sparta %d rows dim %d: min %.3f ms median %.3f ms' % (n, dim, min(ts) * 1e3, np.median(ts) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    M.compute_psf_from_sparta(mk(), **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
