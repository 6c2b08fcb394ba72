"""Host-side cost of queueing reconstruct calls (no GPU wait inside the loop)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muse_psfr_amd import Context, synthetic_rows, grid_pixscale
n = 100
see, gl, l0 = synthetic_rows(n)
lb = np.linspace(465, 930, 35)
three = np.zeros(n, np.uint8)
dev = torch.device('cuda:0')
fits = [torch.zeros((n, 35, 16), dtype=torch.float64, device=dev) for _ in range(2)]
psums = [torch.zeros((35, 40, 40), dtype=torch.float64, device=dev) for _ in range(2)]
for streams in (1, 2):
    ctx = Context(dim=512, pixscale=grid_pixscale(512))
    ctx.set_option('streams', streams)
    for i in range(3):
        ctx.reconstruct_device(lb, see, gl, l0, three, (100, 10000), 12.0, 1, None, None,
                               psums[i % 2].data_ptr(), fits[i % 2].data_ptr())
    ctx.sync()
    for k in (1, 4, 20, 50):
        ctx.profile_reset()
        t0 = time.perf_counter()
        for i in range(k):
            ctx.reconstruct_device(lb, see, gl, l0, three, (100, 10000), 12.0, 1, None, None,
                                   psums[i % 2].data_ptr(), fits[i % 2].data_ptr())
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        hs, hn = ctx.host_time()
        print('streams=%d calls=%2d  enqueue %.3f ms/call (in library %.3f)   total %.3f ms/call' % (
            streams, k, (t1 - t0) / k * 1e3, hs / hn * 1e3, (t2 - t0) / k * 1e3))
    ctx.close()
