"""Builds nerfail_amd/lib/libnerfail_hip.so from nerfail_amd/csrc/*.hip with hipcc for gfx950.

In-tree on purpose: the .so travels to the GPU box with the repository snapshot (it is git-ignored,
not gpurun-ignored). hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container. Usage: `python -m nerfail_amd.build [--force] [--verbose]`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libnerfail_hip.so')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# -ffp-contract=off: the oracle (numpy) has no FMA; kernels that must round like the reference use
# explicit __fmul_rn/__fadd_rn anyway, this keeps the remaining expressions from being contracted.
CFLAGS = ['-O3', '--offload-arch=' + ARCH, '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall',
          '-Wno-unused-function', '-I', INCLUDE]


# per-file flags. mlp_lds.hip: its fully unrolled layer loops (32 quads x 32 pinned MFMAs, some as inline asm) exceed the
# size up to which hipcc honours '#pragma unroll' (16 384); not unrolled, the accumulator arrays would be indexed at run time
# and live in scratch.
FILE_FLAGS = {'mlp_lds.hip': ['-mllvm', '-pragma-unroll-threshold=65536']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(INCLUDE, 'nerfail_hip.h'))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + CFLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        logs = list(ex.map(run, jobs))
    if verbose:
        for lg in logs:
            if lg.strip():
                print(lg)
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    path = build(force='--force' in sys.argv, verbose='--verbose' in sys.argv)
    print(path)
