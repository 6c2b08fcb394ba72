"""Iteration counts of the Moffat fit on the bench workload; saves the slow stamps for analysis."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
n = 100
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
see, gl, l0 = synthetic_rows(n)
lb = np.linspace(465, 930, 35)
ctx = Context(dim=dim, pixscale=grid_pixscale(dim))
r = ctx.reconstruct(lb, see, gl, l0, np.zeros(n, np.uint8), (100, 10000))
ctx.close()
it = r['fit'][:, :, 7]
print('iterations: mean %.2f max %d hist %s' % (it.mean(), it.max(), np.bincount(it.astype(int).ravel())))
slow = np.argwhere(it >= (7 if dim >= 512 else 20))
for t, l in slow[:40]:
    f = r['fit'][t, l]
    print('task %d lam %d it %d status %d fwhm_px %.3f n %.3f peak %.5f chi2 %.3e' % (t, l, f[7], f[14], f[5], f[4], f[0], f[6]))
os.makedirs('gpurun_out/r2', exist_ok=True)
np.savez_compressed('gpurun_out/r2/slow_stamps_%d.npz' % dim, idx=slow, psf=np.array([r['psf'][t, l] for t, l in slow]),
                    fit=np.array([r['fit'][t, l] for t, l in slow]))
