#!/usr/bin/env python3
"""Reads the phase stamps of the `lds_clock` experiment build (tools/experiment.py lds_clock): shader cycles one wave
spends per tile in encode / layer 0 / the D layers / views + heads, median over all waves of the grid."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_%s.so' % (sys.argv[1] if len(sys.argv) > 1 else 'lds_clock'))
import synth  # noqa: E402
from nerfail_amd.run_nerf import _mlp_points  # noqa: E402
from nerfail_amd.run_nerf_helpers import NeRF  # noqa: E402

dev = torch.device('cuda:0')
sd = synth.nerf_state_dict(seed=1)
m = NeRF(8, 256, 63, 27, 5, [4], True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev)
R, N = 8192, 192
pts = torch.randn((R, N, 3), device=dev)
vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
for rep in range(3):
    raw = _mlp_points(m, pts, vd)
torch.cuda.synchronize()
w = raw.reshape(-1).view(torch.int32)[:1024 * 8].reshape(1024, 8).cpu().numpy().astype(np.int64) & 0xffffffff
names = ['encode + park', 'layer 0 (256 MFMA)', '8 layers + alpha (8448 MFMA)', 'views + rgb (576 MFMA)']
mf = [0, 256, 8448, 576]
tot = 0
for i, n in enumerate(names):
    c = float(np.median(w[:, i]))
    tot += c
    print('%-32s %9.0f cycles%s' % (n, c, ('   = %.1f per MFMA (64 = pipe rate)' % (c / mf[i])) if mf[i] else ''))
print('%-32s %9.0f cycles (9280 MFMA x 64 = 593920)' % ('tile', tot))
