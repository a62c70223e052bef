"""K5 alone: nerfail_composite on 640 000 rays x {64, 192} samples, back to back, HIP events (round 4: is the scan's 4.6 TB/s a
property of the kernel or of running right behind the MLP kernel?). NERFAIL_COMPOSITE_KERNEL=1 selects the one-ray-per-wave form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from nerfail_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
R = 640000
for N in (64, 192):
    raw = torch.randn((R, N, 4), device=dev)
    z = torch.sort(torch.rand((R, N), device=dev) * 4 + 2, -1)[0].contiguous()
    rays = torch.randn((R, 11), device=dev)
    o = [torch.empty(s, device=dev) for s in ((R, 3), (R,), (R,), (R, N), (R,))]
    pm = torch.empty((R, 3), device=dev)
    def call():
        _lib.check(lib.nerfail_composite(_lib.dev(raw), _lib.dev(z), _lib.dev(rays), None, R, N, 1, *[_lib.dev(t) for t in o], None, _lib.dev(pm), _lib.stream()))
    for _ in range(3):
        call()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    b = R * (24 * N + 36)
    print('N=%d: %.4f ms per call, %.2f TB/s algorithmic (%.3f of 8 TB/s)' % (N, best, b / best / 1e9, b / best / 1e9 / 8), flush=True)
