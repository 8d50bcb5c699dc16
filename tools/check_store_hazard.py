#!/usr/bin/env python3
"""Static check of the built gfx950 code objects for the wide-store data hazard (profiles/r06_notes.md section 10).

    python tools/check_store_hazard.py [ukbb_cardiac_amd/libukbb_fcn.so] [wait states, default 2]

A vector-memory STORE of more than 8 bytes (buffer_/global_/flat_/scratch_store_dwordx3 / x4) reads its data registers over several
cycles after issue.  A VALU instruction that WRITES one of those registers in the next issue slots corrupts the stored value when the
memory pipeline pushes back.  hipcc pads the hazard except
for MUBUF stores whose soffset is an SGPR, where LLVM assumes it cannot happen; kernels_ws.hip met it on MI355X in r06.

For every such store in every kernel the script walks the following instructions until WAIT wait states have passed (s_nop N counts N + 1,
any other instruction 1; a branch or label ends the window conservatively as a violation only if a write was seen) and reports a VALU /
MFMA write that overlaps the store's data registers.  Exit status 1 if any is found.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
WAIT = 2                                                                 # wait states LLVM pads for the constant-soffset form on gfx940+

STORE = re.compile(r'^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\b')
VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
AREG = re.compile(r'\ba(\d+)\b|\ba\[(\d+):(\d+)\]')


def regs(tok, rx=VREG):
    out = set()
    for m in rx.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def written(op, args):
    """VGPRs a VALU / MFMA instruction writes (its first operand; both operands of the swaps)."""
    if not args:
        return set()
    if op.startswith(('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane')):
        return set()
    if op.startswith('v_'):
        w = regs(args[0])
        if op.startswith(('v_swap', 'v_permlane32_swap', 'v_permlane16_swap')) and len(args) > 1:
            w |= regs(args[1])
        return w
    # a later vector-memory LOAD into the same registers is not the hazard: it queues behind the store in the same unit and its data
    # returns after the store's has been read; LDS returns are ordered by lgkmcnt against the NEXT reader, they do take >= 64 clocks
    return set()


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix='ukbb_isa_')
    try:
        local = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, capture_output=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if 'gfx950' in f)
        if not objs:
            sys.exit('check_store_hazard: no gfx950 code object in %s' % lib)
        for f in objs:
            text = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', os.path.join(tmp, f)],
                                  check=True, capture_output=True, text=True).stdout
            yield f, text
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan(text, WAIT=WAIT):
    """-> (stores seen, list of (kernel, store line, offending line, distance))."""
    bad, n_store, kernel = [], 0, '?'
    lines = text.split('\n')
    insts = []                                                           # (kernel, op, args, raw) ; None marks a label / function boundary
    for l in lines:
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', l)
        if m:
            if not m.group(1).startswith('L'):                           # local labels are LBB..., kernels are mangled names
                kernel = m.group(1)
            insts.append(None)
            continue
        l = l.split('//')[0].strip()
        if not l or l.startswith(('Disassembly', '/')) or ':' in l.split()[0]:
            continue
        parts = l.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
        insts.append((kernel, op, args, l))
    for i, ins in enumerate(insts):
        if ins is None or not STORE.match(ins[1]):
            continue
        n_store += 1
        k, op, args, raw = ins
        # data operand: buffer_store vdata, vaddr, srsrc, soffset ; global_store vaddr, vdata, saddr ; flat_store vaddr, vdata ; scratch_store vaddr, vdata, saddr
        data = regs(args[0]) if op.startswith('buffer') else regs(args[1])
        data_a = regs(args[0], AREG) if op.startswith('buffer') else regs(args[1], AREG)
        waited, j = 0, i + 1
        while waited < WAIT and j < len(insts):
            nx = insts[j]
            if nx is None:                                               # label: a branch target, the window continues on the fall-through path
                j += 1
                continue
            _, nop, nargs, nraw = nx
            if nop == 's_nop':
                waited += int(nargs[0], 0) + 1
            else:
                w = written(nop, nargs)
                if (w & data) or (nop.startswith('v_accvgpr_write') and regs(nargs[0], AREG) & data_a):
                    bad.append((k, raw, nraw, waited))
                    break
                if nop.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc')):
                    break
                waited += 1
            j += 1
    return n_store, bad


def main(lib, WAIT=WAIT):
    total, found = 0, []
    for name, text in disassemble(lib):
        n, bad = scan(text, WAIT)
        total += n
        found += bad
    print('check_store_hazard: %d stores of more than 8 bytes in %s, %d followed by a write of their data registers within %d wait states'
          % (total, os.path.basename(lib), len(found), WAIT))
    demangle = os.path.join(LLVM, 'llvm-cxxfilt')
    for k, st, nx, d in found[:40]:
        try:
            k = subprocess.run([demangle, k], capture_output=True, text=True).stdout.strip()[:120]
        except OSError:
            pass
        print('  %s\n      %s\n      %s   <- %d wait state(s) later' % (k, st, nx, d))
    return 1 if found else 0


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'ukbb_cardiac_amd', 'libukbb_fcn.so'),
                  int(sys.argv[2]) if len(sys.argv) > 2 else WAIT))
