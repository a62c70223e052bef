#!/usr/bin/env python3
"""K6 nerfail_sample_fine at one whole 800 x 800 view per launch (640 000 rays, 64 + 128 samples): ms per launch and GB/s
of its algorithmic traffic (z_vals + weights in, z_fine + pts out = 3.6 KB per ray), ordered u (perturb = 0) and random u."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
R, nc, nf = 640000, 64, 128
g = torch.Generator(device=dev); g.manual_seed(0)
rays = torch.randn((R, 11), device=dev, generator=g)
zc = torch.sort(torch.rand((R, nc), device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
w = torch.rand((R, nc), device=dev, generator=g) ** 8
zf, pts, zstd = torch.empty((R, nc + nf), device=dev), torch.empty((R, nc + nf, 3), device=dev), torch.empty((R,), device=dev)
for name, u, row in (('ordered u (row)', torch.linspace(0, 1, nf, device=dev), 1), ('random u', torch.rand((R, nf), device=dev, generator=g), 0)):
    def run():
        _lib.check(lib.nerfail_sample_fine(_lib.dev(rays), R, _lib.dev(zc), _lib.dev(w), nc, _lib.dev(u), row, nf, None,
                                           _lib.dev(zf), _lib.dev(pts), _lib.dev(zstd), _lib.stream()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = R * 4 * (nc * 2 + (nc + nf) * 4 + (0 if row else nf) + 1)
    print('%-16s %.3f ms  %.2f TB/s (%.2f GB algorithmic)' % (name, ms, by / ms / 1e9, by / 1e9))
