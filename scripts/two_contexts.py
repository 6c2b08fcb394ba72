"""Experiment: two contexts (own streams and workspaces) fed alternately -- how much would
cross-call overlap buy over one context?"""
import sys, os, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale, NFIT
n = 100
see, gl, l0 = synthetic_rows(n)
lb = np.linspace(465, 930, 35)
three = np.zeros(n, np.uint8)
dev = torch.device('cuda:0')
gc.disable()
for nctx in (1, 2, 3, 4):
    ctxs = [Context(dim=512, pixscale=grid_pixscale(512)) for _ in range(nctx)]
    fits = [torch.zeros((n, 35, NFIT), dtype=torch.float64, device=dev) for _ in range(nctx)]
    psums = [torch.zeros((35, 40, 40), dtype=torch.float64, device=dev) for _ in range(nctx)]
    def run(k):
        for i in range(k):
            j = i % nctx
            ctxs[j].reconstruct_device(lb, see, gl, l0, three, (100, 10000), 12.0, 1, None, None,
                                       psums[j].data_ptr(), fits[j].data_ptr())
        for c in ctxs:
            c.sync()
    if len(sys.argv) > 1:
        for c in ctxs:
            c.set_option('profile_only', c.profile_names().index('otf_rowfft'))
            c.set_option('profile', 1)
    run(90)
    for rep in range(3):
        t0 = time.perf_counter()
        K = int(os.environ.get('K', 120))
        run(K)
        dt = (time.perf_counter() - t0) / K
        print('contexts=%d  %.4f ms/call  %.2f M PSFs/s' % (nctx, dt * 1e3, n * 35 / dt / 1e6))
    for c in ctxs:
        c.close()
