"""Phase times of the thin-wave matrix-core kernel (otf_mfma2.hip), one launch of the bench workload.
Needs a library with the clock compiled in:
    python scripts/variants.py build clock=-DMPSFR_MF_CLOCK=1                     (here)
    MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py [permax]         (on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, grid_pixscale
from muse_psfr_amd.synthetic import synthetic_rows
dim, rows, nl = int(os.environ.get('MF_DIM', 512)), 100, 35
see, gl, l0 = synthetic_rows(rows)
lb = np.linspace(465, 930, nl)
ctx = Context(dim=dim, pixscale=grid_pixscale(dim), precision='mixed')
ctx.set_option('streams', 1)
ctx.set_option('mf_clock', 1)
if len(sys.argv) > 1:
    ctx.set_option('mf_permax', int(sys.argv[1]))
for key in ('mf_floor', 'mf_mid_log2'):
    if os.environ.get('MPSFR_' + key.upper()):
        ctx.set_option(key, float(os.environ['MPSFR_' + key.upper()]))
for _ in range(3):
    r = ctx.reconstruct(lb, see, gl, l0, np.zeros(rows, np.uint8), (100, 10000), npsflin=1)
c = ctx.debug_fetch('mf_clock', (4096, 16, 16))
ok = c[:, :, 0] > 0
w = c[ok]
t0 = w[:, 0].min()
names = ['begin', 'masks(1st sweep)', 'k-loops', 'stage() issue', 'tile steps', 'wait loads', 'wait barrier',
         'second pass', 'end', 'k-steps', 'tiles', 'reduce+epilogue', 'dma instrs']
print('waves %d  workgroups %d  kernel span %.0f cycles' % (ok.sum(), ok.any(axis=1).sum(), w[:, 8].max() - t0))
tot = w[:, 8] - w[:, 0]
print('wave lifetime: mean %.0f  p50 %.0f  max %.0f' % (tot.mean(), np.median(tot), tot.max()))
for i in (1, 2, 3, 4, 5, 6, 7, 11):
    print('  %-18s mean %8.0f  max %8.0f   (%.1f %% of the mean lifetime)' % (names[i], w[:, i].mean(), w[:, i].max(), 100 * w[:, i].mean() / tot.mean()))
print('  first stage of an item (issue + wait + barrier): mean %.0f (%.1f %% of the mean lifetime), items per wave %.1f' % (w[:, 14].mean(), 100 * w[:, 14].mean() / tot.mean(), w[:, 13].mean()))
print('  per wave: k-steps %.1f  tile steps %.1f  dma instructions %.1f' % (w[:, 9].mean(), w[:, 10].mean(), w[:, 12].mean()))
print('  per k-step: stage %.0f  tiles %.0f  wait loads %.0f  barrier %.0f' % tuple(w[:, i].sum() / w[:, 9].sum() for i in (3, 4, 5, 6)))
print('  per tile step (cycles of the wave): %.0f' % (w[:, 4].sum() / max(w[:, 10].sum(), 1)))
wg_start = np.array([c[i][ok[i]][:, 0].min() for i in range(c.shape[0]) if ok[i].any()]) - t0
wg_end = np.array([c[i][ok[i]][:, 8].max() for i in range(c.shape[0]) if ok[i].any()]) - t0
print('workgroup start: p50 %.0f p90 %.0f max %.0f | duration mean %.0f max %.0f' % (
    *np.percentile(wg_start, [50, 90, 100]), (wg_end - wg_start).mean(), (wg_end - wg_start).max()))
ctx.close()
