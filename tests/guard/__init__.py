"""Guard-page mode for the GPU tests and bench.py (VERDICT r3 missing 5): NERFAIL_GUARD_ALLOC=1 replaces torch's caching
allocator by tests/guard/guard_alloc.cpp - every tensor ends at the end of its own mapping with unmapped addresses behind
it and is unmapped when freed, so a one-past-the-end access or a pointer-of-a-temporary launch faults deterministically.
Combine with NERFAIL_TRACE=2 (library: name + sync after every launch) to get the kernel's name."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'guard_alloc.cpp')
LIB = os.path.join(HERE, 'libnf_guard_alloc.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run([HIPCC, '-O2', '-shared', '-fPIC', '-std=c++17', SRC, '-o', LIB], check=True)
    return LIB


def wanted():
    return os.environ.get('NERFAIL_GUARD_ALLOC', '0') == '1'


_installed = False


def install():
    """Must run before the first device allocation of the process."""
    global _installed
    if _installed:
        return
    import torch
    if not os.path.exists(LIB):
        raise RuntimeError('NERFAIL_GUARD_ALLOC=1 but %s is not built (python -c "import __graft_entry__ as g; g.build()")' % LIB)
    alloc = torch.cuda.memory.CUDAPluggableAllocator(LIB, 'nf_guard_malloc', 'nf_guard_free')
    torch.cuda.memory.change_current_allocator(alloc)
    _installed = True


def install_if_wanted():
    if wanted():
        install()
        return True
    return False
