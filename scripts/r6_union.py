"""Round 6, VERDICT r5 #2b: which blocks of the half plane does ANY wavelength of a task keep?  Per task: the union of
the matrix-core stage's block masks over the wavelengths, against the support of the telescope OTF.
usage: python scripts/r6_union.py [dim] [rows]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import muse_psfr_amd as M
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100
see, gl, l0 = M.synthetic_rows(rows)
lb = np.linspace(465, 930, 35)
ps = M.grid_pixscale(dim) if dim != 1280 else 0.2
c = M.Context(dim=dim, pixscale=ps)
c.set_option('chunk_tasks', 1)
fr = []
for t in range(rows):
    c.reconstruct(lb, see[t:t + 1], gl[t:t + 1], l0[t:t + 1], want_psf=False)
    w = c.debug_fetch('mf_work', (7,))
    fr.append((w[0] / w[2], w[5] * 35 / w[2], w[6] * 35 / w[2]))
fr = np.array(fr)
print('dim %d, %d rows: executed (mean over wavelengths) %.3f; kept by any wavelength: mean %.3f min %.3f max %.3f; '
      'inside the telescope support %.3f' % (dim, rows, fr[:, 0].mean(), fr[:, 1].mean(), fr[:, 1].min(), fr[:, 1].max(), fr[0, 2]))
print('histogram of the union fraction per task:', np.histogram(fr[:, 1], bins=[0, .1, .2, .3, .4, .5, .6, .7, .8, 1.0])[0].tolist())
order = np.argsort(fr[:, 1])
for t in list(order[:3]) + list(order[-3:]):
    print('  seeing %.2f GL %.2f L0 %.1f: union %.3f' % (see[t], gl[t], l0[t], fr[t, 1]))
