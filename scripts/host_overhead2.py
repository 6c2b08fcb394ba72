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
fit = torch.zeros((n, 35, 16), dtype=torch.float64, device=dev)
psum = torch.zeros((35, 40, 40), dtype=torch.float64, device=dev)
ctx = Context(dim=512, pixscale=grid_pixscale(512))
ctx.set_option('streams', int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(3):
    ctx.reconstruct_device(lb, see, gl, l0, three, (100, 10000), 12.0, 1, None, None, psum.data_ptr(), fit.data_ptr())
ctx.sync()
K = 400
ts = np.zeros(K + 1)
ts[0] = time.perf_counter()
for i in range(K):
    ctx.reconstruct_device(lb, see, gl, l0, three, (100, 10000), 12.0, 1, None, None, psum.data_ptr(), fit.data_ptr())
    ts[i + 1] = time.perf_counter()
ctx.sync()
tend = time.perf_counter()
d = np.diff(ts) * 1e3
print('total %.3f ms/call; per-call host ms: median %.3f mean %.3f p90 %.3f max %.3f' % ((tend - ts[0]) / K * 1e3, np.median(d), d.mean(), np.percentile(d, 90), d.max()))
for a in range(0, K, 50):
    print('calls %3d-%3d mean %.3f' % (a, a + 49, d[a:a + 50].mean()))
