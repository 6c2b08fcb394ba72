"""Fraction of (task, line) pairs K_DPHI_SERIES_Q skips on the bench rows (stage_a_queue = 2).  usage: [dim] [rows]"""
import sys, numpy as np
sys.path.insert(0, '.')
import muse_psfr_amd as M
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100
see, gl, l0 = M.synthetic_rows(rows)
lb = np.linspace(465 if dim != 1280 else 490, 930, 35)
ps = M.grid_pixscale(dim) if dim != 1280 else 0.2
c = M.Context(dim=dim, pixscale=ps)
c.set_option('stage_a_queue', 2)
c.set_option('chunk_tasks', 25)
tel = None
fr = []
for t0 in range(0, rows, 25):
    sl = slice(t0, t0 + 25)
    c.reconstruct(lb, see[sl], gl[sl], l0[sl], want_psf=False)
    d = c.debug_fetch('dphi0', (25, 1, dim // 2 + 1, dim))[:, 0]
    if tel is None:
        tel = c.debug_fetch('tel', (dim // 2 + 1, dim)) > 0
    sk = ((d >= 1e29) | ~tel[None]).all(axis=2) & tel.any(axis=1)[None]
    fr.append(sk.mean(axis=1))
fr = np.concatenate(fr)
print('dim %d, %d bench rows: lines skipped: mean %.3f; per task: %d rows skip nothing, %d rows more than half; max %.3f' % (
    dim, rows, fr.mean(), (fr == 0).sum(), (fr > 0.5).sum(), fr.max()))
