"""The built gfx950 code objects hold no 16-byte store whose data registers a VALU instruction overwrites in the next two issue slots.

hipcc pads that hazard itself except for buffer stores whose soffset is an SGPR; kernels_ws.hip met the unpadded form on MI355X in round 6 (wrong tiles
of the bf16 U-Net whenever another stream's kernels ran beside it; profiles/r06_notes.md section 10) and pads by hand (store_b128_sofs).  This is
the CPU-side guard against the next kernel that writes such a store: tools/check_store_hazard.py disassembles every code object of the library."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _tool():
    import check_store_hazard as c
    if not os.path.exists(os.path.join(c.LLVM, 'llvm-objdump')):
        pytest.skip('no llvm-objdump in this image')
    return c


def test_scanner_finds_the_pattern_it_is_looking_for():
    c = _tool()
    text = '''
0000000000001000 <_Zkernel>:
	buffer_store_dwordx4 v[128:131], v0, s[12:15], s55 offen   // 000000001000: E07C1000 37038000
	v_cvt_pk_bf16_f32 v128, v26, v27                           // 000000001008: D2680080 0002371A
	buffer_store_dwordx4 v[4:7], v0, s[12:15], s55 offen
	s_nop 1
	v_mov_b32_e32 v5, 0
	global_store_dwordx4 v[2:3], v[8:11], off
	v_add_f32_e32 v20, v8, v9
	v_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]
	buffer_store_dwordx2 v[30:31], v0, s[12:15], s55 offen
	v_mov_b32_e32 v30, 0
	global_store_dwordx4 v[2:3], v[40:43], off
	v_add_f32_e32 v20, v8, v9
	v_mov_b32_e32 v43, 0
'''
    n, bad = c.scan(text)
    assert n == 4                                                          # the 8-byte store is not one
    assert [(b[1].split()[0], b[2].split()[0], b[3]) for b in bad] == [('buffer_store_dwordx4', 'v_cvt_pk_bf16_f32', 0), ('global_store_dwordx4', 'v_mfma_f32_32x32x2_f32', 1),
                                                                       ('global_store_dwordx4', 'v_mov_b32_e32', 1)]


def test_library_has_no_unpadded_wide_store():
    c = _tool()
    lib = os.path.join(ROOT, 'ukbb_cardiac_amd', 'libukbb_fcn.so')
    assert os.path.exists(lib), 'build the library first (__graft_entry__.build())'
    total, found = 0, []
    for _, text in c.disassemble(lib):
        n, bad = c.scan(text)
        total += n
        found += bad
    assert total > 1000                                                    # the scan saw the code objects (2808 such stores in the r06 build)
    assert not found, found[:5]
