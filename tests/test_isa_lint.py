"""Static hazard check of the emitted gfx950 ISA (tools/isa_lint.py): the safety net behind the
hand-scheduled inline asm of the matrix-core kernels.  Runs without a GPU: hipcc cross-compiles,
llvm-objdump disassembles."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_lint as L  # noqa: E402


def _kernel(body):
    """objdump-style text of one kernel from 'op operands' lines (addresses 4 bytes apart)."""
    out = ['0000000000001000 <k>:']
    for i, ln in enumerate(body):
        out.append('\t%-58s // %012X: 00000000' % (ln, 0x1000 + 4 * i))
    return '\n'.join(out) + '\n'


def _rules(body):
    return sorted({r[0] for r in L.lint_text(_kernel(body))})


DMA_OK = ['s_nop 4', 's_mov_b32 s5, m0', 's_mov_b32 m0, s19', 's_nop 0',
          'global_load_lds_dwordx4 v8, s[8:9]', 's_mov_b32 m0, s5']


def test_rules_on_handwritten_snippets():
    # the statement as the library writes it, right behind a v_readfirstlane of its base: clean
    assert _rules(['v_readfirstlane_b32 s8, v1', 'v_readfirstlane_b32 s9, v2'] + DMA_OK + ['s_endpgm']) == []
    # the recorded bug: opening pad of one wait state only
    bad = ['v_readfirstlane_b32 s8, v1', 's_nop 0'] + DMA_OK[1:] + ['s_endpgm']
    assert _rules(bad) == ['R1', 'R1s']
    # ... also when the writer reaches the load through a branch
    via_branch = ['v_readlane_b32 s9, v127, 3', 's_branch 1', 's_endpgm'] + DMA_OK[2:] + ['s_endpgm']
    assert 'R1' in _rules(via_branch)
    # a vector instruction that does not write the base: only the strict rule
    assert _rules(['v_add_u32_e32 v3, v1, v2', 's_nop 0'] + DMA_OK[1:] + ['s_endpgm']) == ['R1s']
    # M0 written right before the LDS-DMA
    assert _rules(['s_nop 4', 's_mov_b32 m0, s19', 'global_load_lds_dwordx4 v8, s[8:9]', 's_endpgm']) == ['R2']
    # vector write -> MFMA operand
    mf = 'v_mfma_f32_16x16x32_f16 v[0:3], v[10:13], v[20:23], v[0:3]'
    assert _rules(['v_cvt_pk_f16_f32 v10, v40, v41', mf, 's_endpgm']) == ['R3']
    assert _rules(['v_cvt_pk_f16_f32 v10, v40, v41', 's_nop 0', mf, 's_endpgm']) == ['R3']
    assert _rules(['v_cvt_pk_f16_f32 v10, v40, v41', 's_nop 1', mf, 's_endpgm']) == []
    assert _rules(['v_cvt_pk_f16_f32 v30, v40, v41', mf, 's_endpgm']) == []
    # an accumulate chain is no hazard (MFMA write -> next MFMA's C)
    assert _rules([mf, mf, 's_endpgm']) == []
    # transcendental -> next vector instruction
    assert _rules(['v_exp_f32_e32 v4, v5', 'v_fma_mix_f32 v6, v7, -1.0, v4 op_sel_hi:[1,0,0]', 's_endpgm']) == ['R4']
    assert _rules(['v_exp_f32_e32 v4, v5', 'v_exp_f32_e32 v8, v9',
                   'v_fma_mix_f32 v6, v7, -1.0, v4 op_sel_hi:[1,0,0]', 's_endpgm']) == []
    # a dependent chain of transcendentals stays in their own pipe: no hazard (hipcc emits these back to back)
    assert _rules(['v_rcp_f32_e32 v4, v9', 'v_exp_f32_e32 v0, v4', 's_nop 0', 'v_add_f32_e32 v1, v0, v0', 's_endpgm']) == []
    assert _rules(['v_rcp_f32_e32 v4, v9', 'v_exp_f32_e32 v0, v4', 'v_add_f32_e32 v1, v0, v0', 's_endpgm']) == ['R4']
    # a barrier behind an LDS-DMA load without the vmcnt(0) wait
    assert _rules(DMA_OK + ['s_waitcnt lgkmcnt(0)', 's_barrier', 's_endpgm']) == ['R5']
    assert _rules(DMA_OK + ['s_waitcnt vmcnt(0) lgkmcnt(0)', 's_barrier', 's_endpgm']) == []


def test_shipped_library_is_clean():
    from muse_psfr_amd._build import build_library
    lib = build_library(force=False, verbose=False)
    findings, nkernels, ninstr = L.lint_library(lib)
    assert nkernels > 50 and ninstr > 50000          # the code objects were found and parsed
    assert findings == [], '\n'.join('%s %s: %s' % f for f in findings[:20])


@pytest.mark.parametrize('src', ['otf_mfma.hip'])
def test_a_build_without_the_opening_pad_is_flagged(src):
    """-DMPSFR_MF_BASE_NOP=0 is the bug of round 2 (scalar operand produced by the vector pipe less
    than five wait states before the LDS-DMA that reads it: stale base, stamps off by 1e-5, no
    fault, and only when the scheduler happened to put the producer next to the statement)."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    tmp = tempfile.mkdtemp(prefix='isa_lint_nop0_')
    try:
        co = os.path.join(tmp, 'nop0.co')
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize',
                        '--offload-device-only', '--no-gpu-bundle-output', '-DMPSFR_MF_BASE_NOP=0',
                        '-x', 'hip', '-c', os.path.join(ROOT, 'muse_psfr_amd', 'csrc', src), '-o', co],
                       check=True, capture_output=True)
        findings = L.lint_text(L.disassemble_code_object(co))
        assert any(f[0] in ('R1', 'R1s') for f in findings)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
