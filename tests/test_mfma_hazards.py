"""The hand-issued VGPR-form MFMAs of the W = 256 inference kernel (nerfail_amd/csrc/mlp_lds.hip, lds_part<.., VG>) are opaque
inline asm to hipcc: it inserts no hazard wait states around them. tools/check_mfma_hazards.py checks the shipped code object
instruction by instruction; here: the checker itself on hand-made listings, then the built library."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_mfma_hazards as H  # noqa: E402

HEAD = '0000000000001000 <kern>:\n'


def _listing(body):
    lines, addr = [], 0x1000
    for ins in body:
        tgt = ''
        if ins.startswith('s_cbranch') or ins.startswith('s_branch'):
            ins, off = ins.rsplit('@', 1)
            tgt = ' <kern+%#x>' % int(off, 0)
        lines.append('\t%-60s // %012X: BF800000%s' % (ins.strip(), addr, tgt))
        addr += 4
    return HEAD + '\n'.join(lines) + '\n'


MF = 'v_mfma_f32_32x32x2_f32 v[2:17], v142, v170, v[2:17]'


def _check(body):
    funcs = H.parse(_listing(body))
    return H.check_function('kern', funcs['kern'])


def test_checker_accepts_accumulate_chains_and_distant_uses():
    n, bad = _check([MF, MF, 'v_mfma_f32_32x32x2_f32 a[0:15], v1, v30, a[0:15]', 's_nop 15', 's_nop 3', 'v_max_i32_e32 v1, 0, v2', 's_endpgm'])
    assert n == 2 and bad == []
    # 19 wait states between the MFMA and the use: the first legal slot
    n, bad = _check([MF] + ['s_nop 0'] * 19 + ['v_mov_b32_e32 v40, v17', 's_endpgm'])
    assert bad == []


def test_checker_flags_every_kind_of_early_touch():
    for early in ('v_max_i32_e32 v1, 0, v2',                                  # VALU read of the result (the next layer's operand)
                  'v_mov_b32_e32 v17, v40',                                   # VALU write (WAW / a copy placed early)
                  'ds_read_b128 v[4:7], v200 offset:1024',                    # LDS return into the array (a bias tile)
                  'scratch_store_dwordx4 off, v[14:17], s32 offset:16',       # a spill
                  'global_store_dword v201, v9, s[4:5]',                      # VMEM read
                  'v_mfma_f32_32x32x2_f32 a[0:15], v142, v2, a[0:15]',        # another MFMA reading it as SrcB
                  'v_mfma_f32_32x32x2_f32 v[10:25], v142, v170, v[10:25]'):   # a partially overlapping accumulator
        n, bad = _check([MF] + ['s_nop 0'] * 18 + [early, 's_endpgm'])
        assert len(bad) == 1 and '18 wait state' in bad[0], (early, bad)
        n, bad = _check([MF, 's_nop 15', 's_nop 2', early, 's_endpgm'])      # s_nop N counts N + 1: 19 wait states in between
        assert bad == []


def test_checker_follows_branches():
    # the touch sits on the taken path of a conditional branch, and behind a backward branch (a loop's next iteration)
    n, bad = _check([MF, 's_cbranch_scc1 9 @0x18', 's_nop 15', 's_nop 3', 's_endpgm', 's_nop 0', 'v_max_i32_e32 v1, 0, v9', 's_endpgm'])
    assert len(bad) == 1 and 'v_max_i32' in bad[0]
    n, bad = _check(['v_max_i32_e32 v1, 0, v9', MF, 's_nop 3', 's_branch 65530 @0x0'])
    assert len(bad) == 1
    n, bad = _check(['v_max_i32_e32 v1, 0, v9', MF, 's_nop 15', 's_nop 3', 's_branch 65530 @0x0'])
    assert bad == []


def test_shipped_library_has_no_vgpr_form_mfma_hazard():
    from nerfail_amd import build as B
    lib = B.build()
    total, bad, where = H.check(lib)
    names = [n for n, _ in where]
    # the three SKIP variants of the W = 256 inference kernel, nothing else (training and backward kernels: all MFMAs builtin)
    assert len(where) == 3 and all('nerf_mlp_fwd_lds_kernelILi8E' in n and n.endswith('Lb0EEEvNS_7MlpArgsE') for n in names), names
    assert total >= 3 * 1280
    assert bad == [], '\n'.join(bad[:20])
