#!/usr/bin/env python3
"""Build-time check of the hand-issued VGPR-form MFMAs (nerfail_amd/csrc/mlp_lds.hip: lds_part<.., VG>).

hipcc picks ONE MFMA form per function (accumulators in AGPRs here: two 128-register activation arrays). The inference kernel
keeps one of the two arrays in arch VGPRs and issues that array's MFMAs as inline asm (`v_mfma_f32_32x32x2_f32 v[..], v, v,
v[..]`). The compiler does not know such a statement is a 16-pass XDL op, so it inserts none of the wait states it would put
between an MFMA and an instruction that touches its result registers. This script disassembles the code object that SHIPS
(the .hip_fatbin of libnerfail_hip.so or of an object file) and asserts, for EVERY v_mfma whose destination is an arch-VGPR
range D:

  within the WINDOW = 19 wait states that follow it (the longest MFMA hazard of gfx940+: XDL write VGPR -> VALU / VMEM / LDS /
  FLAT read or write, and XDL write -> XDL read as SrcA / SrcB, for a 16-pass op; wait states are counted the way hipcc's
  hazard recogniser counts them: one per instruction, s_nop N = N + 1), along EVERY control-flow path,
  no instruction touches a register of D, except a v_mfma that accumulates into exactly D (SrcC = vDst = D, back to back:
  the hardware forwards it; this is how every VGPR-form GEMM loop runs).

Anything else - a register copy, a spill, a v_max that prepares the next layer's operand, a ds_read that writes a bias tile,
a partially overlapping MFMA - is a violation and fails the build (`__graft_entry__.build()` and tests/test_abi.py run this).

    python3 tools/check_mfma_hazards.py [path/to/libnerfail_hip.so | object.o] [--verbose]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
WINDOW = 19

_INS = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)(?:<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>)?')
_FUNC = re.compile(r'^([0-9a-f]+) <([^>]+)>:')
_VREG = re.compile(r'\bv(?:\[(\d+):(\d+)\]|(\d+))')


def disassemble(path):
    """Text disassembly of every gfx950 code object bundled in `path` (a .so or .o built by hipcc)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            raise RuntimeError('no .hip_fatbin section in %s' % path)
        # a linked .so carries one bundle per translation unit, each 4096-aligned: unbundle them one by one
        blob = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        out = []
        for i, s in enumerate(starts):
            part = os.path.join(tmp, 'b%d.bin' % i)
            with open(part, 'wb') as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                                '--targets=' + TARGET, '--output=' + co], capture_output=True, text=True)
            if r.returncode != 0:
                continue
            out.append(subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], check=True,
                                      capture_output=True, text=True).stdout)
        if not out:
            raise RuntimeError('no %s bundle in %s' % (TARGET, path))
        return '\n'.join(out)


def vregs(text):
    s = set()
    for m in _VREG.finditer(text):
        if m.group(3) is not None:
            s.add(int(m.group(3)))
        else:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


def parse(dis):
    """-> {function: [(addr, mnemonic, operands, branch target address or None)]}"""
    funcs, cur, base = {}, None, {}
    for line in dis.splitlines():
        f = _FUNC.match(line)
        if f:
            cur = f.group(2)
            base[cur] = int(f.group(1), 16)
            funcs[cur] = []
            continue
        m = _INS.match(line)
        if not m or cur is None:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith('s_cbranch') or mnem == 's_branch':
            if m.group(5) is None:
                raise RuntimeError('branch without a resolved target: ' + line)
            tgt = base.get(m.group(5), None)
            if tgt is None:
                raise RuntimeError('branch into another symbol: ' + line)
            tgt += int(m.group(6) or '0', 16)
        funcs[cur].append((addr, mnem, ops, tgt))
    return funcs


def wait_states(mnem, ops):
    if mnem == 's_nop':
        return int(ops.split()[0], 0) + 1
    return 1


def check_function(name, ins):
    """Violations of the rule in the module docstring, as text lines. `ins` = parse()'s list for one function."""
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    bad, n_vg = [], 0
    for i, (addr, mnem, ops, _) in enumerate(ins):
        if not mnem.startswith('v_mfma') or not ops.startswith('v['):
            continue
        n_vg += 1
        parts = [p.strip() for p in ops.split(',')]
        dst = vregs(parts[0])
        if vregs(parts[3]) != dst:
            bad.append('%s %#x: VGPR-form MFMA whose SrcC is not its vDst: %s %s' % (name, addr, mnem, ops))
        # walk every path for WINDOW wait states
        work, seen = [(i + 1, 0)], set()
        while work:
            k, ws = work.pop()
            while k < len(ins) and ws < WINDOW:
                if (k, ws) in seen:
                    break
                seen.add((k, ws))
                a2, m2, o2, tgt = ins[k]
                if m2 == 's_endpgm':
                    break
                touched = vregs(o2) & dst
                if touched:
                    ok = False
                    if m2.startswith('v_mfma'):
                        p2 = [p.strip() for p in o2.split(',')]
                        ok = (vregs(p2[0]) == dst and vregs(p2[3]) == dst and not (vregs(p2[1]) | vregs(p2[2])) & dst)
                    if not ok:
                        bad.append('%s: %#x `%s %s` touches v%s %d wait state(s) after the MFMA at %#x (`%s`): needs >= %d'
                                   % (name, a2, m2, o2, sorted(touched)[:4], ws, addr, ops, WINDOW))
                        break
                ws += wait_states(m2, o2)
                if tgt is not None:
                    if tgt not in index:
                        bad.append('%s: branch at %#x leaves the function' % (name, a2))
                    else:
                        work.append((index[tgt], ws))
                    if m2 == 's_branch':
                        break
                k += 1
    return n_vg, bad


def check(path, verbose=False):
    funcs = parse(disassemble(path))
    total, bad, where = 0, [], []
    for name, ins in funcs.items():
        n, b = check_function(name, ins)
        total += n
        bad += b
        if n:
            where.append((name, n))
    if verbose:
        for name, n in where:
            print('%6d VGPR-form MFMAs in %s' % (n, name))
    return total, bad, where


def main(argv):
    verbose = '--verbose' in argv
    args = [a for a in argv if not a.startswith('--')]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = args[0] if args else os.path.join(here, 'nerfail_amd', 'lib', 'libnerfail_hip.so')
    total, bad, where = check(path, verbose)
    for b in bad[:40]:
        print('HAZARD', b)
    print('%s: %d VGPR-form MFMAs in %d kernels, %d hazard violation(s) within %d wait states'
          % (os.path.basename(path), total, len(where), len(bad), WINDOW))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
