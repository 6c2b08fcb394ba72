#!/usr/bin/env python3
"""Static check of the gfx950 code objects of libmpsfr.so for the hazards that hipcc does NOT pad
when one side of them sits inside an inline-asm statement ("Wait states inside
asm strings are the author's"; /opt/skills/guides/cdna_hip_programming.md section 5.7).

The library's kernels issue LDS-DMA loads (global_load_lds_dwordx4), v_fma_mix_f32 and -- in
experiments -- v_pk_fma_f32 through inline asm.  The compiler schedules such a statement as one
opaque instruction: it neither pads the wait states its operands need nor counts its memory
operations.  Two such hazards produced silently wrong stamps in round 2 and passed every test until
an unrelated edit moved instructions.  This tool disassembles the emitted ISA and checks, for EVERY
instruction of every kernel (it cannot tell compiler instructions from asm ones, and does not need to):

  R1  an SGPR written by a VECTOR instruction (v_readfirstlane, v_readlane, v_cmp, carry-outs)
      is not read by a vector-memory instruction (base, soffset, descriptor) within 5 wait states;
  R1s the same for an LDS-DMA load (which only inline asm issues, with scalar operands the compiler
      may have produced any way it likes, a v_readlane of a spilled SGPR included), in the strict
      form the statements are written to: NO vector instruction at all in the 5 wait states
      before it on any path -- so that the statement is safe wherever the scheduler puts it, not
      only where it happens to stand in this build (the opening s_nop 4 of every load statement);
  R2  M0 is not written in the wait state before an LDS-DMA instruction that uses it;
  R3  a VGPR written by a (non-MFMA) vector instruction is not read by a v_mfma_* as A, B or C
      within 2 wait states;
  R4  a VGPR written by a transcendental instruction (v_exp, v_log, v_rcp, v_rsq, v_sqrt, v_sin,
      v_cos) is not read by the next vector instruction (1 wait state) -- unless that instruction is
      itself transcendental: the hazard is the forwarding from the transcendental unit to the main
      vector pipe (LLVM's GCNHazardRecognizer checks exactly this for gfx940+: a dependent chain of
      transcendentals stays in their in-order pipe, and hipcc emits such chains back to back);
  R5  in a kernel that issues LDS-DMA loads, every s_barrier is preceded, on every path, by an
      s_waitcnt vmcnt(0) with no LDS-DMA load in between (a barrier does not drain the DMA, and the
      waves behind it read what the DMA wrote).

  R6  a VGPR written by a vector instruction is not read through DPP (the first source of a *_dpp
      instruction: the row_newbcast operands of stage_a2.hip) within 2 wait states, and EXEC written
      by a vector instruction is not followed by a DPP instruction within 5.

Wait states are counted conservatively: every instruction is one, s_nop N is N + 1; a distance is
taken over every path that reaches the instruction (fall-through and branches).

    python tools/isa_lint.py muse_psfr_amd/libmpsfr.so        # exit code 1 if anything is flagged
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')
INSTR_RE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):')
LABEL_RE = re.compile(r'^([0-9A-Fa-f]+) <(.+)>:')
REG_RE = re.compile(r'\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|vcc_lo|vcc_hi|m0|exec|exec_lo|exec_hi)\b')


def regs_of(text):
    """Set of registers named in an operand string: ('v', 3), ('s', 8), ('vcc',), ('m0',)."""
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
        else:
            out.add((m.group(6).split('_')[0],))
    return out


class Ins:
    __slots__ = ('addr', 'op', 'operands', 'ops')

    def __init__(self, addr, op, operands):
        self.addr, self.op, self.operands = addr, op, operands
        self.ops = [o.strip() for o in operands.split(',')] if operands else []

    def __repr__(self):
        return '%x: %s %s' % (self.addr, self.op, self.operands)

    # ---- classification
    def is_valu(self):
        return self.op.startswith('v_') and not self.op.startswith('v_mfma') and not self.op.startswith('v_smfmac')

    def is_mfma(self):
        return self.op.startswith('v_mfma') or self.op.startswith('v_smfmac')

    def is_trans(self):
        return self.op.startswith(TRANS)

    def is_vmem(self):
        return self.op.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))

    def is_lds_dma(self):
        return self.op.startswith('global_load_lds') or (self.op.startswith('buffer_load') and ' lds' in ' ' + self.operands)

    def wait_states(self):
        if self.op == 's_nop':
            return int(self.ops[0], 0) + 1
        return 1

    def dst_regs(self):
        """Registers written (vector instructions and the scalar ones that matter here)."""
        if not self.ops:
            return set()
        d = set()
        if self.is_valu() or self.is_mfma():
            if self.op.startswith('v_cmpx'):
                d.add(('exec',))
            elif self.op.startswith('v_cmp'):
                d |= regs_of(self.ops[0]) if self.op.endswith('_e64') or self.ops[0].startswith(('s', 'vcc')) else {('vcc',)}
                if self.op.endswith('_e32') or not (self.ops[0].startswith(('s', 'vcc'))):
                    d.add(('vcc',))
            else:
                d |= regs_of(self.ops[0])
                if '_co_' in self.op or self.op.startswith(('v_div_scale', 'v_mad_u64', 'v_mad_i64')):
                    if len(self.ops) > 1 and self.ops[1].startswith(('s', 'vcc')):
                        d |= regs_of(self.ops[1])
                    else:
                        d.add(('vcc',))
        elif self.op.startswith('s_') and not self.op.startswith(('s_cmp', 's_bitcmp', 's_waitcnt', 's_nop',
                                                                   's_branch', 's_cbranch', 's_barrier',
                                                                   's_endpgm', 's_setprio', 's_sleep')):
            d |= regs_of(self.ops[0])
        return d

    def src_regs(self):
        if self.is_valu() or self.is_mfma():
            first = 1
            if self.op.startswith('v_cmp') and not (self.ops and self.ops[0].startswith(('s', 'vcc'))):
                first = 0
            s = set()
            for o in self.ops[first:]:
                s |= regs_of(o)
            # read-modify-write destinations (v_fmac, v_dot2c, v_pk_fmac, v_mac ...)
            if re.match(r'v_(pk_)?(fmac|mac|dot\dc)', self.op):
                s |= regs_of(self.ops[0])
            return s
        s = set()
        for o in self.ops[(0 if self.is_vmem() and 'store' in self.op else 0):]:
            s |= regs_of(o)
        return s


def parse_kernels(text):
    """{symbol: [Ins, ...]} from `llvm-objdump -d` output."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = LABEL_RE.match(line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = INSTR_RE.match(line)
        if m and cur is not None:
            cur.append(Ins(int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels


def branch_target(ins):
    if ins.op.startswith(('s_branch', 's_cbranch')) and ins.ops:
        try:
            off = int(ins.ops[0], 0)
        except ValueError:
            return None
        if off >= 1 << 15:
            off -= 1 << 16
        return ins.addr + 4 + 4 * off
    return None


def predecessors(code):
    idx = {ins.addr: i for i, ins in enumerate(code)}
    preds = [[] for _ in code]
    for i, ins in enumerate(code):
        t = branch_target(ins)
        if t is not None and t in idx:
            preds[idx[t]].append(i)
        falls = not (ins.op in ('s_branch', 's_endpgm') or ins.op.startswith('s_setpc'))
        if falls and i + 1 < len(code):
            preds[i + 1].append(i)
    return preds


def walk_back(code, preds, i, budget, visit):
    """Call visit(j, states_between) for every instruction j that can execute fewer than `budget`
    wait states before instruction i (states_between = wait states issued strictly between j and i).
    visit returns True to stop going further back along that path."""
    stack = [(p, 0) for p in preds[i]]
    seen = set()
    while stack:
        j, between = stack.pop()
        if (j, between) in seen or between >= budget:
            continue
        seen.add((j, between))
        if visit(j, between):
            continue
        nb = between + code[j].wait_states()
        for p in preds[j]:
            stack.append((p, nb))


def lint_kernel(name, code):
    out = []
    preds = predecessors(code)
    has_dma = any(ins.is_lds_dma() for ins in code)
    for i, ins in enumerate(code):
        if ins.is_vmem():
            sregs = {r for r in ins.src_regs() if r[0] in ('s', 'vcc')}
            if sregs:
                def v1(j, between, sregs=sregs, i=i):
                    pj = code[j]
                    if pj.is_valu() and (pj.dst_regs() & sregs):
                        out.append(('R1', name, '%r reads %s written by %r %d wait state(s) earlier (need 5)' % (
                            code[i], sorted(pj.dst_regs() & sregs), pj, between)))
                        return True
                    return bool(pj.dst_regs() & sregs) and False
                walk_back(code, preds, i, 5, v1)
        if ins.is_lds_dma():
            def v1s(j, between, i=i):
                pj = code[j]
                if pj.is_valu() or pj.is_mfma():
                    out.append(('R1s', name, '%r: vector instruction %r %d wait state(s) before an LDS-DMA load '
                                             '(its statement must open with s_nop 4)' % (code[i], pj, between)))
                    return True
                return False
            walk_back(code, preds, i, 5, v1s)

            def v2(j, between, i=i):
                pj = code[j]
                if ('m0',) in pj.dst_regs():
                    out.append(('R2', name, '%r uses M0 written by %r in the previous wait state' % (code[i], pj)))
                return True
            walk_back(code, preds, i, 1, v2)
        if ins.is_mfma():
            vregs = {r for o in ins.ops[1:4] for r in regs_of(o) if r[0] in ('v', 'a')}
            def v3(j, between, vregs=vregs, i=i):
                pj = code[j]
                if pj.is_valu() and (pj.dst_regs() & vregs):
                    out.append(('R3', name, '%r reads %s written by %r %d wait state(s) earlier (need 2)' % (
                        code[i], sorted(pj.dst_regs() & vregs), pj, between)))
                    return True
                return False
            walk_back(code, preds, i, 2, v3)
        if (ins.is_valu() and not ins.is_trans()) or ins.is_mfma():
            vsrc = {r for r in ins.src_regs() if r[0] == 'v'}
            def v4(j, between, vsrc=vsrc, i=i):
                pj = code[j]
                if pj.is_trans() and (pj.dst_regs() & vsrc):
                    out.append(('R4', name, '%r reads %s written by %r in the previous wait state' % (
                        code[i], sorted(pj.dst_regs() & vsrc), pj)))
                return True
            walk_back(code, preds, i, 1, v4)
        if ins.is_valu() and ins.op.endswith('_dpp') and len(ins.ops) > 1:
            dsrc = {r for r in regs_of(ins.ops[1]) if r[0] == 'v'}
            def v6(j, between, dsrc=dsrc, i=i):
                pj = code[j]
                if pj.is_valu() and (pj.dst_regs() & dsrc):
                    out.append(('R6', name, '%r reads %s through DPP, written by %r %d wait state(s) earlier (need 2)' % (
                        code[i], sorted(pj.dst_regs() & dsrc), pj, between)))
                    return True
                return False
            walk_back(code, preds, i, 2, v6)
            def v6x(j, between, i=i):
                pj = code[j]
                if pj.is_valu() and ('exec',) in pj.dst_regs():
                    out.append(('R6', name, '%r: EXEC written by %r %d wait state(s) before a DPP instruction (need 5)' % (
                        code[i], pj, between)))
                    return True
                return False
            walk_back(code, preds, i, 5, v6x)
        if has_dma and ins.op == 's_barrier':
            # every path backwards must meet an s_waitcnt vmcnt(0) before it meets an LDS-DMA load
            stack, seen, bad = list(preds[i]), set(), None
            while stack and bad is None:
                j = stack.pop()
                if j in seen:
                    continue
                seen.add(j)
                pj = code[j]
                if pj.op == 's_waitcnt' and re.search(r'vmcnt\(0\)', pj.operands):
                    continue
                if pj.is_lds_dma():
                    bad = pj
                    break
                stack.extend(preds[j])
            if bad is not None:
                out.append(('R5', name, '%r can be reached from %r without an s_waitcnt vmcnt(0)' % (ins, bad)))
    return out


def lint_text(text):
    res = []
    for name, code in parse_kernels(text).items():
        if code:
            res += lint_kernel(name, code)
    return res


def disassemble_code_object(path):
    return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', path], check=True,
                          capture_output=True, text=True).stdout


def code_objects_of(lib):
    """Extract the gfx950 code objects bundled in a host shared library / object into a temp dir."""
    tmp = tempfile.mkdtemp(prefix='isa_lint_')
    local = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True,
                   capture_output=True, text=True, cwd=tmp)
    cos = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if 'amdgcn' in f and 'gfx950' in f)
    return tmp, cos


def lint_library(lib):
    tmp, cos = code_objects_of(lib)
    try:
        if not cos:
            raise RuntimeError('no gfx950 code object found in %s' % lib)
        res, nk, ni = [], 0, 0
        for co in cos:
            text = disassemble_code_object(co)
            ks = parse_kernels(text)
            nk += len(ks)
            ni += sum(len(c) for c in ks.values())
            res += lint_text(text)
        return res, nk, ni
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'muse_psfr_amd', 'libmpsfr.so')
    res, nk, ni = lint_library(lib)
    for rule, kern, msg in res:
        print('%s %s: %s' % (rule, kern[:60], msg))
    print('%d symbols, %d instructions, %d finding(s)' % (nk, ni, len(res)))
    sys.exit(1 if res else 0)
