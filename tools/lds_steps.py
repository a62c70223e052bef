#!/usr/bin/env python3
"""Reads the per-step stamps of the `lds_steps` experiment build: shader cycles of each of the last 64 steps (16 MFMAs
each; 1024 cycles = pipe rate) of a wave's final tile, median over the waves of the grid."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_%s.so' % (sys.argv[1] if len(sys.argv) > 1 else 'lds_steps'))
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
w = raw.reshape(-1).view(torch.int32)[:1024 * 66].reshape(1024, 66).cpu().numpy().astype(np.int64) & 0xffffffff
idx = int(w[0, 64])
print('steps per wave:', idx, '(580 per tile)')
lo = int(sys.argv[2]) if len(sys.argv) > 2 else idx - 64      # first recorded step (the build's NF_ST_LO)
order = [(lo + k) % 64 for k in range(64)]             # oldest .. newest
t = w[:, order]
d = np.diff(t, axis=1) & 0xffffffff
med = np.median(d, axis=0)
first = lo
print(' '.join('%d:%.0f' % ((first + 1 + k) % 580, v) for k, v in enumerate(med)))
print('mean of the 63 steps: %.0f cycles (1024 = 16 MFMAs at pipe rate)' % med.mean())
