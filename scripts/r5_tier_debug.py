import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muse_psfr_amd as M
H = (100, 10000)
dim, ps = 512, M.grid_pixscale(512)
see = np.array([0.6, 0.5, 0.3, 0.9, 0.4]); gl = np.array([0.5, 0.9, 0.5, 0.9, 0.95]); l0 = np.array([15.0, 20.0, 29.0, 11.0, 29.0])
lb = np.linspace(465, 930, 7)
out = {}
for key, opts in (('no_tiers', {'tier_eps': 0}), ('default', {}), ('tight', {'tier_eps': 1e-8}), ('tight_nomid', {'tier_eps': 1e-8, 'mf_mid_log2': -1e30}),
                  ('tight_nofloor', {'tier_eps': 1e-8, 'mf_floor_log2': -1e30}), ('floor29', {'mf_floor_log2': -29.01})):
    ctx = M.Context(dim=dim, pixscale=ps, precision='mixed')
    for k, v in opts.items():
        ctx.set_option(k, v)
    r = ctx.reconstruct(lb, see, gl, l0, None, H)
    out[key] = ctx.debug_fetch('pre', (5, 7, 40, 40))
    out[key + '_w'] = ctx.debug_fetch('mf_work', (5,))
    ctx.close()
ref = out['no_tiers']
peak = ref.max(axis=(2, 3)); q = 1600 * peak / ref.sum(axis=(2, 3))
np.set_printoptions(linewidth=200, precision=2)
print('1600 peak/sum'); print(q)
for key in ('default', 'tight', 'tight_nomid', 'tight_nofloor', 'floor29'):
    err = np.abs(out[key] - ref).max(axis=(2, 3)) / peak
    print(key, 'work', out[key + '_w'], 'no_tiers work', out['no_tiers_w']); print(err)
