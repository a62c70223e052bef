#!/usr/bin/env python3
"""Where a wave of the gauss row-reduce kernel spends its life (needs the NF_ROW_ABLATE=9 build: every lane writes the
wall-clock ticks of header / staging / row walk instead of its result)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['NERFAIL_HIP_LIB'] = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_NF_ROW_ABLATE_9.so')
import synth
from nerfail_amd.GaussNet import gauss_gather, create_gauss_w
dev = torch.device('cuda:0')
H = W = 800
P, B = 3, 8
rs = np.random.RandomState(0)
Ns = P * H * W
base = (rs.randint(0, P, size=(B, 1, 1, 1)) * H * W + np.arange(H * W).reshape(1, H, W, 1))
idx = np.clip(base + rs.randint(-2 * W, 2 * W, size=(B, H, W, 8)), 0, Ns - 1).astype(np.float32)
dist_ = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
wi, _ = create_gauss_w(dev, 0.02)(torch.from_numpy(np.stack([dist_, idx], 1)).to(dev))
ori = torch.from_numpy(synth.disc_alpha_image(B, H, W, seed=3)).to(dev)
G = torch.from_numpy(rs.normal(size=(B, H, W, 4)).astype(np.float32)).to(dev)
for rep in range(2):
    s = torch.zeros((P, H, W, 4), device=dev)
    s[..., 3] = 255.0
    st = s.requires_grad_(True)
    x, xr = gauss_gather(st, wi, ori, None, None, True)
    xr.backward(G)
    torch.cuda.synchronize()
t = st.grad.reshape(-1, 4).cpu().numpy()
print('per lane, mean ticks of 10 ns: header %.0f  staging %.0f  row walk %.0f   (entries per wave: mean %.0f, max %.0f)' %
      (t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), t[:, 3].max()))
print('wave lifetime ~ %.1f us; 30000 waves at 2048 resident -> %.2f ms' % (t[:, :3].sum(1).mean() * 1e-2, t[:, :3].sum(1).mean() * 1e-5 * 30000 / 2048))
