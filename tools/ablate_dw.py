#!/usr/bin/env python3
"""Timing ablations of the LDS-staged weight-gradient kernel: builds libnerfail_hip_ablateN.so (N = NF_DW_ABLATE) next
to the product library. `python tools/ablate_dw.py build` here, then on the GPU
`python tools/microbench_mlp.py --lib nerfail_amd/lib/libnerfail_hip_ablateN.so --only bwd_w_bf16x3`."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfail_amd import build as B

def main():
    B.build()
    for v in (1, 2, 3):
        o = os.path.join(B.OBJDIR, 'mlp_bwd_ablate%d.o' % v)
        subprocess.check_call([B.HIPCC] + B.CFLAGS + ['-DNF_DW_ABLATE=%d' % v, '-c', os.path.join(B.CSRC, 'mlp_bwd.hip'), '-o', o])
        objs = [os.path.join(B.OBJDIR, f[:-4] + '.o') for f in B._sources() if f != 'mlp_bwd.hip'] + [o]
        lib = os.path.join(B.LIBDIR, 'libnerfail_hip_ablate%d.so' % v)
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs)
        print(lib)

if __name__ == '__main__':
    main()
