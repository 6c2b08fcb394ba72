"""In-tree build of libmpsfr.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the tree to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmpsfr.so')
SOURCES = ['stage_a.hip', 'stage_a2.hip', 'per_lambda.hip', 'otf_mfma.hip', 'otf_mfma2.hip', 'stamps.hip', 'mpsfr_api.cpp']
HEADERS = ['kernels.h', 'fft_lds.h', 'fft_r16.h', 'device_common.h', 'mf_common.h', 'coeff_l0_table.h', 'psd_model.h', 'dpp_groups.h', 'conv_frames.h',
           os.path.join('..', '..', 'include', 'mpsfr.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-result',
         '-fno-slp-vectorize']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libmpsfr.so')
    return exe


STAMP = LIB + '.stamp'


def source_hash():
    """sha256 over the sources, headers and compiler flags: the identity of a build.  It is
    written beside the .so (libmpsfr.so.stamp) and compiled into it (mpsfr_build_id)."""
    h = hashlib.sha256()
    h.update(' '.join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def is_stale():
    """A shipped .so is only reused if it was built from exactly these sources and flags."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build_library(force=False, verbose=True, out=None, extra_flags=()):
    """Compile csrc/*.hip, csrc/*.cpp into muse_psfr_amd/libmpsfr.so.  Returns the path.
    `out` / `extra_flags`: build a variant (e.g. -DMPSFR_...=...) into another file, for kernel
    experiments (load it with MPSFR_LIB_PATH)."""
    variant = out is not None
    if not variant and not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    build_id = source_hash() + ('+' + ' '.join(extra_flags) if variant else '')
    objs = []
    bdir = os.path.join(HERE, 'build' + ('_' + os.path.basename(out) if variant else ''))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:          # the translation units compile in parallel
        obj = os.path.join(bdir, os.path.splitext(src)[0] + '.o')
        cmd = [hipcc] + FLAGS + list(extra_flags) + ['-DMPSFR_BUILD_ID="%s"' % build_id, '-x', 'hip', '-c',
                                 os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    target = out if variant else LIB
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', target] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    if not variant:
        lint_library(target)
        with open(STAMP, 'w') as fh:
            fh.write(build_id + '\n')
    return target


def lint_library(lib):
    """Static hazard check of the emitted ISA (tools/isa_lint.py): hipcc pads nothing inside an
    inline-asm statement, so a build whose instruction placement exposes a hazard must not ship.
    MPSFR_SKIP_ISA_LINT=1 skips it (kernel experiments)."""
    if os.environ.get('MPSFR_SKIP_ISA_LINT'):
        return
    import importlib.util
    tool = os.path.join(os.path.dirname(HERE), 'tools', 'isa_lint.py')
    if not os.path.exists(tool):
        return
    spec = importlib.util.spec_from_file_location('isa_lint', tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    findings, _, _ = mod.lint_library(lib)
    if findings:
        os.remove(lib)
        raise RuntimeError('ISA hazard lint failed for %s:\n%s' % (
            lib, '\n'.join('%s %s: %s' % f for f in findings[:20])))


if __name__ == '__main__':
    print(build_library(force=True))
