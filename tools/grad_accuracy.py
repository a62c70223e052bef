"""True accuracy of the MLP parameter gradients: HIP f32 kernels, HIP split-precision kernels and torch fp32 autograd,
each against a float64 torch evaluation of the same network on the same points and the same upstream gradient d_raw
(captured from a real 1024-ray training step so the conditioning is the real one).

    python tools/grad_accuracy.py            (needs the GPU)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import synth                                     # noqa: E402
from hiputil import hip_nerf, T, torch_nerf_mlp, hip_mlp_grads   # noqa: E402
from nerfail_amd import _train, run_nerf as RN   # noqa: E402


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.manual_seed(7)
    rays = T(synth.ray_batch(1024, seed=5))
    target = torch.rand((1024, 3), device=rays.device)
    sd, coarse = hip_nerf(8, 256, 31, requires_grad=True)
    _, fine = hip_nerf(8, 256, 32, requires_grad=True)
    cap = []
    orig_bwd2, orig_fwd_rays = _train.mlp_backward2, _train.mlp_fwd_train_rays

    def fwd_rays(net, rays_, z_vals, acts=None):          # the training pipeline forms the sample points inside the kernel
        cap.append(('f', net, (rays_[:, None, 0:3] + rays_[:, None, 3:6] * z_vals[..., None]).clone(), rays_[:, 8:11].clone()))
        return orig_fwd_rays(net, rays_, z_vals, acts)

    def bwd2(net0, d_raw, acts, grads0, M0, net1, grads1, M1, accumulate=False):       # coarse + fine in one launch
        cap.append(('b', net0, d_raw[:M0].clone()))
        return orig_bwd2(net0, d_raw, acts, grads0, M0, net1, grads1, M1, accumulate)
    _train.mlp_fwd_train_rays, _train.mlp_backward2 = fwd_rays, bwd2
    r = RN.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.)
    (RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)).backward()
    _train.mlp_fwd_train_rays, _train.mlp_backward2 = orig_fwd_rays, orig_bwd2
    pts, dirs = [(c[2], c[3]) for c in cap if c[0] == 'f' and c[1] is coarse][0]
    d_raw = [c[2] for c in cap if c[0] == 'b' and c[1] is coarse][0].reshape(pts.shape[0], pts.shape[1], 4)
    R, N = pts.shape[0], pts.shape[1]
    flat_dirs = dirs[:, None, :].expand(R, N, 3).reshape(-1, 3) if dirs.dim() == 2 else dirs.reshape(-1, 3)

    def torch_grads(dtype):
        raw, P = torch_nerf_mlp(sd, pts.reshape(-1, 3), flat_dirs, dtype)
        (raw * d_raw.reshape(-1, 4).to(dtype)).sum().backward()
        return {k: v.grad.double().cpu().numpy() for k, v in P.items()}

    def hip_grads(fwd, bd, dw):
        return hip_mlp_grads(coarse, pts, dirs, d_raw, fwd, bd, dw)
    truth = torch_grads(torch.float64)
    cand = {'torch_f32': torch_grads(torch.float32), 'hip_f32': hip_grads('f32', 'f32', 'f32'),
            'split_all': hip_grads('split', 'split', 'split'), 'split_fwd': hip_grads('split', 'f32', 'f32'),
            'split_bd': hip_grads('f32', 'split', 'f32'), 'split_dw': hip_grads('f32', 'f32', 'split')}
    print(('%-26s' + ' %10s' * len(cand)) % ('parameter', *cand))
    for k in truth:
        n = np.linalg.norm(truth[k])
        print(('%-26s' + ' %10.2e' * len(cand)) % ((k,) + tuple(np.linalg.norm(c[k] - truth[k]) / n for c in cand.values())))


if __name__ == '__main__':
    main()
