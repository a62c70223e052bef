#!/usr/bin/env python3
"""Timing ablations: builds libnerfail_hip_<macro>_<value>.so next to the product library with one source recompiled
under -D<macro>=<value>.  `python tools/ablate.py mlp.hip NF_FWD_ABLATE 1 2 3` here, then on the GPU
`python tools/microbench_mlp.py --lib nerfail_amd/lib/libnerfail_hip_NF_FWD_ABLATE_1.so --only fwd_infer`.
(The variant libraries are git-ignored; delete them afterwards, they travel with every gpurun snapshot.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfail_amd import build as B


def main():
    src, macro, values = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    for v in values:
        o = os.path.join(B.OBJDIR, '%s_%s_%s.o' % (src[:-4], macro, v))
        subprocess.check_call([B.HIPCC] + B.CFLAGS + ['-D%s=%s' % (macro, v), '-c', os.path.join(B.CSRC, src), '-o', o])
        objs = [os.path.join(B.OBJDIR, f[:-4] + '.o') for f in B._sources() if f != src] + [o]
        lib = os.path.join(B.LIBDIR, 'libnerfail_hip_%s_%s.so' % (macro, v))
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs)
        print(lib)


if __name__ == '__main__':
    main()
