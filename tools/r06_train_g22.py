#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 1: train a D=8 W=256 coarse + fine pair (the headline shape, RN:435-441) with the PRODUCT on the
GPU box and write the two state dicts to gpurun_out/g22_weights.npz. They are INPUT DATA of fixture g22: in the build container
tests/golden/make_golden.py g22 hands them to the REFERENCE, which renders 4 096 rays (deterministic + perturbed, fp32 + fp64)
and computes one training step's gradients (fp32 + fp64) with them. The scene, the ray set, the schedule and the draws are
those of g21 / tests/test_hip_f16x3.py::test_f16x3_on_trained_like_weights (analytic unit sphere, 40 poses of 100 x 100 rays,
1 024 rays per step, perturb = 1, Adam lr 5e-4 with the reference's decay, RN:776-801).

    python3 tools/r06_train_g22.py [steps]      (on the GPU box; ~15 s for 2 000 steps)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402

SEED_COARSE, SEED_FINE = 221, 222


def sphere_target(rays):
    o, d = rays[:, 0:3].astype(np.float64), rays[:, 3:6].astype(np.float64)
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    b = (o * dn).sum(1)
    disc = b * b - ((o * o).sum(1) - 1.0)
    t = -b - np.sqrt(np.maximum(disc, 0.0))
    hit = (disc > 0) & (t > 0)
    nrm = o + dn * t[:, None]
    return np.where(hit[:, None], 0.5 + 0.5 * nrm, 1.0).astype(np.float32), hit


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    from nerfail_amd import run_nerf as RN
    from nerfail_amd.optim import Adam
    from nerfail_amd.run_nerf import ray_gen
    from nerfail_amd.run_nerf_helpers import NeRF
    dev = torch.device('cuda:0')

    def net(seed):
        sd = synth.nerf_state_dict(D=8, W=256, seed=seed)
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev)
    coarse, fine = net(SEED_COARSE), net(SEED_FINE)
    opt = Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    Hs = Ws = 100
    focal, K = synth.lego_intrinsics(Hs, Ws)
    rays_all = torch.cat([ray_gen(Hs, Ws, K, synth.pose_spherical(float(th), -30., 4.)[:3, :4], 2., 6.)
                          for th in np.linspace(-180, 180, 41)[:-1]])
    tgt_np, hit = sphere_target(rays_all.cpu().numpy())
    tgt_all = torch.from_numpy(tgt_np).to(dev)
    gen = torch.Generator(device=dev).manual_seed(22)
    t0, first = time.time(), None
    for it in range(steps):
        sel = torch.randint(0, rays_all.shape[0], (1024,), device=dev, generator=gen)
        r = RN.render_rays(rays_all[sel].contiguous(), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True,
                           perturb=1., t_rand=torch.rand((1024, 64), device=dev, generator=gen),
                           u=torch.rand((1024, 128), device=dev, generator=gen))
        loss = RN.img2mse(r['rgb_map'], tgt_all[sel]) + RN.img2mse(r['rgb0'], tgt_all[sel])
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g['lr'] = 5e-4 * 0.1 ** ((it + 1) / 500000.)
        if it == 0:
            first = float(loss.detach())
        if it % 200 == 0:
            print('g22 training step %4d loss %.5f (%.1f s)' % (it, float(loss.detach()), time.time() - t0), flush=True)
    last = float(loss.detach())
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    out = {'steps': steps, 'loss_first': first, 'loss_last': last, 'seed_coarse': SEED_COARSE, 'seed_fine': SEED_FINE,
           'device': torch.cuda.get_device_name(0)}
    for nm, m in (('coarse', coarse), ('fine', fine)):
        for k, v in m.state_dict().items():
            out['%s_%s' % (nm, k)] = v.detach().cpu().numpy()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    path = os.path.join(ROOT, 'gpurun_out', 'g22_weights.npz')
    np.savez_compressed(path, **out)
    print('g22: loss %.5f -> %.5f in %d steps (%.1f s); hit fraction %.3f; wrote %s (%.1f MB)'
          % (first, last, steps, time.time() - t0, float(hit.mean()), path, os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    main()
