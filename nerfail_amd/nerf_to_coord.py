"""MI355X mirror of Create_spatial_point_set/nerf_to_coord.py (reference = NC): the render-only copy of
run_nerf.py that additionally returns `pts_max`, the 3-D sample point with the largest fine weight per
pixel (NC:418-423), and saves it as {i:03d}.npy [H,W,3] float32 (NC:172-173)."""
import os
import time

import numpy as np

from . import run_nerf as _rn
from .run_nerf import (batchify, run_network, create_nerf, raw2outputs, FusedNetworkQuery, ray_gen)  # noqa: F401
from .run_nerf_helpers import to8b


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                **extra):
    """NC:320-436: run_nerf.render_rays plus ret['pts_max'] (argmax of the final weights, first maximum)."""
    return _rn.render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=retraw, lindisp=lindisp,
                           perturb=perturb, N_importance=N_importance, network_fine=network_fine,
                           white_bkgd=white_bkgd, raw_noise_std=raw_noise_std, verbose=verbose, pytest=pytest,
                           want_pts_max=True, **extra)


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """NC:55-67."""
    return _rn.batchify_rays(rays_flat, chunk, want_pts_max=True, **kwargs)


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """NC:70-135 -> [rgb_map, disp_map, acc_map, pts_max, extras]."""
    return _rn._render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, True, kwargs)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """NC:138-178: per view render + (optional) PNG and the pts_max .npy the 8-NN build consumes."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    t = time.time()
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        rgb, disp, acc, pts_max, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if savedir is not None:
            try:
                import imageio
                imageio.imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
            except ImportError:
                pass
            np.save(os.path.join(savedir, '{:03d}.npy'.format(i)), pts_max.cpu().numpy())
    return np.stack(rgbs, 0), np.stack(disps, 0)
