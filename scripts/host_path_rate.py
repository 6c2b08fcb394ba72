"""PCIe-inclusive rates of the host-buffer boundary (mpsfr_reconstruct with on_device = 0): inputs
and outputs in host memory, one synchronous call per step."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
n = 100
see, gl, l0 = synthetic_rows(n)
lb = np.linspace(465, 930, 35)
three = np.zeros(n, np.uint8)
ctx = Context(dim=512, pixscale=grid_pixscale(512))
for want_psf in (False, True):
    for _ in range(5):
        ctx.reconstruct(lb, see, gl, l0, three, (100, 10000), want_psf=want_psf)
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        r = ctx.reconstruct(lb, see, gl, l0, three, (100, 10000), want_psf=want_psf)
    dt = (time.perf_counter() - t0) / K
    print('host outputs: fit table + stamp sum%s: %.3f ms per call, %.2f M PSFs/s' % (
        ' + all stamps (45 MB D2H)' if want_psf else '', dt * 1e3, n * 35 / dt / 1e6))
ctx.close()
