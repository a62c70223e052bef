#!/usr/bin/env python3
"""Where does a training step's wall time go? Segment timings with synchronisation (diagnostic only)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import synth
from nerfail_amd import run_nerf as RN, _train
from nerfail_amd.run_nerf_helpers import NeRF
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
def net(seed):
    sd = synth.nerf_state_dict(seed=seed)
    m = NeRF(8, 256, 63, 27, 5, [4], True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.precision = prec
    return m.to(dev)
coarse, fine = net(31), net(32)
params = list(coarse.parameters()) + list(fine.parameters())
from nerfail_amd.optim import Adam
opt = Adam(params, lr=5e-4)
rays = torch.from_numpy(synth.ray_batch(1024, seed=1)).to(dev)
target = torch.rand((1024, 3), device=dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
T = {}
for it in range(12):
    t_rand = torch.rand((1024, 64), device=dev); u = torch.rand((1024, 128), device=dev)
    t0 = sync()
    for n in (coarse, fine):
        n.packed()
        if prec == 'f16x3':
            n.packed_f16(); _train.packed_f16_T(n)
        else:
            _train.packed_T(n)
    t1 = sync()
    r = RN.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1., t_rand=t_rand, u=u)
    loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
    t2 = sync()
    opt.zero_grad()
    t3 = sync()
    loss.backward()
    t4 = sync()
    opt.step()
    t5 = sync()
    if it >= 2:
        for k, v in (('pack', t1 - t0), ('forward', t2 - t1), ('zero_grad', t3 - t2), ('backward', t4 - t3), ('adam', t5 - t4)):
            T.setdefault(k, []).append(v * 1e3)
print(prec, {k: round(float(np.median(v)), 3) for k, v in T.items()}, 'sum', round(sum(float(np.median(v)) for v in T.values()), 3))
# free-running steps (no synchronisation inside): the real step time, to compare with the kernel-time sum of a
# `rocprofv3 --kernel-trace --stats` run of this script
NFREE = 20
t0 = sync()
for it in range(NFREE):
    t_rand = torch.rand((1024, 64), device=dev); u = torch.rand((1024, 128), device=dev)
    r = RN.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1., t_rand=t_rand, u=u)
    loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
    opt.zero_grad()
    loss.backward()
    opt.step()
t_host = time.perf_counter()
t1 = sync()
print('free-running: %.3f ms/step wall, host enqueue %.3f ms/step' % ((t1 - t0) / NFREE * 1e3, (t_host - t0) / NFREE * 1e3))
