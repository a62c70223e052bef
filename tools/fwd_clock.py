#!/usr/bin/env python3
"""Shader clock actually held during the exact-f32 forward kernel (needs `python tools/experiment.py fwd_clock`,
which makes every workgroup write its clock64 / wall_clock64 deltas over the first outputs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_fwd_clock.so')
import synth
from nerfail_amd.run_nerf_helpers import NeRF
dev = torch.device('cuda:0')
sd = synth.nerf_state_dict(seed=1)
m = NeRF(8, 256, 63, 27, 5, [4], True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev).requires_grad_(False)
R, N = 8192, 192
pts = torch.randn((R, N, 3), device=dev)
vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
raw = torch.empty((R, N, 4), device=dev)
lib = _lib.load()
lib.nerfail_mlp_fwd_select(1)          # the probe lives in the register-streamed kernel
for _ in range(3):
    _lib.check(lib.nerfail_mlp_fwd(_lib.dev(m.packed()), m.D, m.W, m._skip(), _lib.dev(pts), _lib.dev(vd), R * N, N, _lib.dev(raw), _lib.stream()))
torch.cuda.synchronize()
q = raw.view(torch.int64).reshape(-1)[:512].cpu().numpy().reshape(256, 2).astype(np.float64)
ghz = q[:, 0] / (q[:, 1] * 10.0)
print('shader clock during nerf_mlp_fwd_kernel: mean %.3f GHz (min %.3f, max %.3f); kernel %.2f ms by wall ticks' % (ghz.mean(), ghz.min(), ghz.max(), q[:, 1].max() * 1e-5))
