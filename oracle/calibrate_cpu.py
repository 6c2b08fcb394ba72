"""CPU-baseline calibration.  TEST INFRASTRUCTURE; runs ONLY in the build container:

    /opt/conda/bin/python3.9 -B oracle/calibrate_cpu.py

bench.py's `cpu_baseline` is the NumPy restatement oracle/psfr_oracle.py ("port"), because the
reference cannot travel to the GPU box.  This script times the REAL reference
(/root/reference/muse_psfr/psfrec.py via oracle/_refload.py: simul_psd_wfm -> psf_muse ->
convolve_final_psf, plus the scipy Moffat fit that stands in for mpdaf on both sides) against
the port on the same rows, in the same interpreter, one process (the reference's n_jobs=1), and
writes the ratio to profiles/r02_cpu_calibration.json.  Configs: P = native 1280^2, 35 lambda
490-930 nm, and the bench workload's 512^2 grid (reference source with its
hard-coded dim/pixscale patched in memory).
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
from _refload import load_reference  # noqa: E402
import psfr_oracle as O  # noqa: E402
from muse_psfr_amd.synthetic import synthetic_rows, grid_pixscale  # noqa: E402

H = (100, 10000)


def time_config(dim, lb, nrows):
    ps = grid_pixscale(dim)
    ref = load_reference() if dim == 1280 else load_reference(dim=dim, pixscale=ps)
    see, gl, l0 = synthetic_rows(nrows)
    t_ref = t_port = 0.0
    worst = 0.0
    for i in range(nrows):
        t = time.perf_counter()
        psd = ref.simul_psd_wfm([gl[i], 1 - gl[i]], H, see[i], l0[i], zenith=0., npsflin=1, dim=dim,
                                three_lgs_mode=False, verbose=False)
        pre = ref.psf_muse(psd[0], lb)
        fin = ref.convolve_final_psf(lb, see[i], gl[i], l0[i], pre)
        fit_r = O.fit_psf_cube(fin, ps)
        t_ref += time.perf_counter() - t
        t = time.perf_counter()
        fit_p, fin_p = O.compute_psf(lb, see[i], gl[i], l0[i], 1, H, False, dim=dim, pixscale=ps)
        t_port += time.perf_counter() - t
        worst = max(worst, float(np.abs(fin_p - fin).max() / fin.max()))
        print('N=%d row %d: reference %.2fs port %.2fs  |stamp diff| %.1e' % (
            dim, i, t_ref, t_port, worst), flush=True)
    n = nrows * lb.size
    return {'dim': dim, 'nl': int(lb.size), 'rows': nrows,
            'reference_psfs_per_s_1core': round(n / t_ref, 3),
            'port_psfs_per_s_1core': round(n / t_port, 3),
            'reference_over_port_time': round(t_ref / t_port, 3),
            'max_rel_stamp_diff_port_vs_reference': worst}


if __name__ == '__main__':
    out = {'what': 'wall time of the real reference (psfrec.py, imported unmodified; N != 1280 via '
                   'its hard-coded dim/pixscale patched in memory) over the NumPy port that '
                   'bench.py times as cpu_baseline, same rows, same interpreter, one process',
           'host': '%s, %d cpus (build container)' % (platform.processor() or platform.machine(),
                                                      os.cpu_count()),
           'python': sys.version.split()[0], 'numpy': np.__version__,
           'configs': [time_config(1280, np.linspace(490, 930, 35), 2),
                       time_config(512, np.linspace(465, 930, 35), 4)]}
    out['reference_over_port_time_512'] = out['configs'][1]['reference_over_port_time']
    out['note'] = ('cpu_baseline.value (port) / reference_over_port_time = what the reference '
                   'itself would deliver on the same cores')
    dst = os.path.join(HERE, '..', 'profiles', 'r02_cpu_calibration.json')
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out, indent=1))
