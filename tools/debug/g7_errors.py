"""Per-parameter error of the HIP training step against the reference's fp32 gradients / the float64-backward oracle, next to
the reference's own fp32-vs-fp64 spread stored in fixture g7 (round 4: choosing the bound of tests/test_hip_train.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
from conftest import l2_err
from hiputil import T, N, hip_nerf
from oracle import nerf as O
from nerfail_amd import run_nerf as RN
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g7_train_grads.npz')))
for prec in ('f32',):
    for tag, D, W in (('small', 4, 64), ('full', 8, 256)):
        sc, coarse = hip_nerf(D, W, 31, requires_grad=True, precision=prec)
        sf, fine = hip_nerf(D, W, 32, requires_grad=True, precision=prec)
        rays, target = g[tag + '_rays'], g[tag + '_target']
        r = RN.render_rays(T(rays), coarse, None, 64, retraw=True, N_importance=128, network_fine=fine, white_bkgd=True,
                           perturb=1., t_rand=T(g[tag + '_t_rand']), u=T(g[tag + '_u']))
        loss = RN.img2mse(r['rgb_map'], T(target)) + RN.img2mse(r['rgb0'], T(target))
        loss.backward()
        ref = O.train_step_grads(rays, sc, sf, target, t_rand=g[tag + '_t_rand'], u=g[tag + '_u'], D=D, W=W)
        print(tag, prec, 'loss', float(loss), float(g[tag + '_loss']), float(g[tag + '_loss64']))
        for nm, net in (('coarse', coarse), ('fine', fine)):
            for k, p in net.named_parameters():
                got = N(p.grad)
                e_or = l2_err(got, ref['grads_' + nm][k])
                e_ref = l2_err(got, g['%s_%s_grad_%s' % (tag, nm, k)])
                e_or_ref = l2_err(ref['grads_' + nm][k], g['%s_%s_grad_%s' % (tag, nm, k)])
                sp = float(g['%s_%s_referr_%s' % (tag, nm, k)])
                print('%-5s %-6s %-24s hip-vs-ref32 %.2e  hip-vs-oracle %.2e  oracle-vs-ref32 %.2e  ref spread %.2e  ratio %.2f'
                      % (tag, nm, k, e_ref, e_or, e_or_ref, sp, e_ref / (2 * sp + (3e-4 if k.startswith('alpha_linear') else 2e-6))))
