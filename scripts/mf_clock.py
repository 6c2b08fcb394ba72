"""Phase time stamps of the matrix-core per-wavelength kernel (one launch, bench workload).
Needs a library built with the clock compiled in:
    python scripts/variants.py build clock=-DMPSFR_MF_CLOCK=1      (here)
    MPSFR_LIB_PATH=variants/clock.so python scripts/mf_clock.py [prune_eps]      (on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, grid_pixscale
from muse_psfr_amd.synthetic import synthetic_rows
dim, rows, nl = int(os.environ.get('MF_DIM', 512)), 100, 35
npl = int(os.environ.get('MF_NPL', 1))
see, gl, l0 = synthetic_rows(rows)
lb = np.linspace(465, 930, nl)
ctx = Context(dim=dim, pixscale=grid_pixscale(dim), precision='mixed')
ctx.set_option('streams', 1)
ctx.set_option('mf_clock', 1)
eps = float(sys.argv[1]) if len(sys.argv) > 1 else None
if eps is not None:
    ctx.set_option('prune_eps', eps)
for _ in range(3):
    r = ctx.reconstruct(lb, see, gl, l0, np.zeros(rows, np.uint8), (100, 10000), npsflin=npl)
nwg = 520
c = ctx.debug_fetch('mf_clock', (nwg, 8, 8))
cc = c[:, :7, :]
c = cc[:, :, :6]
ok = c[:, :, 0] > 0
t0 = c[ok][:, 0].min()
d = np.diff(c, axis=2)[ok]
print('waves', ok.sum(), ' kernel span (cycles)', c[ok][:, 5].max() - t0)
for i, name in enumerate(('masks', 'first stage+barrier', 'k-loop', 'second pass', 'epilogue')):
    print('%-22s mean %8.0f  max %8.0f' % (name, d[:, i].mean(), d[:, i].max()))
print('all groups: k-loops mean %.0f max %.0f | second passes mean %.0f max %.0f' % (cc[ok][:, 6].mean(), cc[ok][:, 6].max(), cc[ok][:, 7].mean(), cc[ok][:, 7].max()))
print('in the k-loops: wait for loads mean %.0f | wait at the barrier mean %.0f | iterations mean %.1f' % (cc[ok][:, 1].mean(), cc[ok][:, 2].mean(), cc[ok][:, 3].mean()))
print('in the k-loops: slab read + staging issue mean %.0f | tile steps mean %.0f' % (cc[ok][:, 0].mean(), cc[ok][:, 4].mean()))
print('start spread: p50 %.0f p90 %.0f max %.0f' % tuple(np.percentile(c[ok][:, 0] - t0, [50, 90, 100])))
print('wave total: mean %.0f max %.0f' % ((c[ok][:, 5] - c[ok][:, 0]).mean(), (c[ok][:, 5] - c[ok][:, 0]).max()))
ctx.close()
