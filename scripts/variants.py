"""Kernel experiments: build variants of libmpsfr.so with extra -D flags (here, in the build
container), then run bench.py against each on the GPU box and print one line per variant.

    python scripts/variants.py build name1=-DFOO=1 name2="-DBAR=2 -DBAZ"     # here
    python scripts/variants.py run [bench args...]                            # on the GPU box
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, 'variants')
sys.path.insert(0, ROOT)


def build(specs):
    from muse_psfr_amd._build import build_library
    os.makedirs(VDIR, exist_ok=True)
    for f in os.listdir(VDIR):
        if f.endswith('.so'):
            os.remove(os.path.join(VDIR, f))
    for spec in specs:
        name, _, flags = spec.partition('=')
        build_library(out=os.path.join(VDIR, name + '.so'), extra_flags=flags.split(), verbose=False)
        print('built', name, flags, flush=True)


def run(args):
    libs = [None] + sorted(f for f in os.listdir(VDIR) if f.endswith('.so'))
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env['MPSFR_LIB_PATH'] = os.path.join(VDIR, lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env,
                           capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        if p.returncode or not line:
            print('%-14s FAILED rc=%d %s' % (lib or 'default', p.returncode, p.stderr[-400:]), flush=True)
            continue
        d = json.loads(line[0])
        par = d.get('parity') or {}
        k = d['kernel_ms_per_step']
        ka = d.get('kernel_ms_one_call_in_flight') or {}
        print('%-14s %.3f M/s  step %.4f ms | otf %.4f prep %.4f fit %.4f colpass %.4f conv %.4f psd %.4f colfft %.4f'
              ' | alone: fit %.4f conv %.4f otf %.4f | dbeta %.1e dfwhm %.1e | it %s' % (
                  lib or 'default', d['value'] / 1e6, d['ms_per_step'], k.get('otf_mfma', 0) + k.get('otf_rowfft', 0),
                  k.get('mf_prep', 0) + k.get('vkeep', 0),
                  k.get('fit', 0), k.get('colpass', 0), k.get('conv', 0), k.get('psd_rowfft', 0),
                  k.get('colfft_dphi', 0), ka.get('fit', 0), ka.get('conv', 0), ka.get('otf_mfma', 0), par.get('max_abs_err_beta', float('nan')),
                  par.get('max_abs_err_fwhm_arcsec', float('nan')), d['fit_iterations']), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
