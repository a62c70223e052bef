"""MI355X mirror of Create_spatial_point_set/nerf_pytorch/run_nerf.py (reference = RN) and of the
render functions of Create_spatial_point_set/nerf_to_coord.py (reference = NC, which adds `pts_max`).

Drop-in surface: batchify, run_network, batchify_rays, render, render_path, create_nerf, raw2outputs,
render_rays keep the reference's signatures and return structures. The arithmetic runs in
libnerfail_hip.so; chunking arguments (`chunk`, `netchunk`) are accepted and honoured as upper bounds
but do not change results. Random draws can be passed explicitly (t_rand=, u=, noise=) so CPU-oracle and
GPU runs share identical seeds; otherwise they are drawn on the device (or from numpy seed 0 when
pytest=True, as RN:374-377 / RH:215-223 / RN:288-291).
"""
import os
import time

import numpy as np
import torch

from . import _lib
from .optim import Adam, decayed_lrate  # noqa: F401
from .run_nerf_helpers import (NeRF, Embedder, get_embedder, get_rays, get_rays_np, img2mse, mse2psnr, to8b,  # noqa: F401
                               sample_pdf, linspace01, _cuda, _k4, _c2w12)

np.random.seed(0)      # RN:23
DEBUG = False


def batchify(fn, chunk):
    """RN:27-34."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def _is_fused(embed_fn, embeddirs_fn, fn):
    return (isinstance(fn, NeRF) and isinstance(embed_fn, Embedder) and embed_fn.multires == 10 and
            isinstance(embeddirs_fn, Embedder) and embeddirs_fn.multires == 4)


def _mlp_points(fn, pts, viewdirs):
    """Fused encode + MLP on raw points: pts [R,N,3], viewdirs [R,3] -> raw [R,N,4] (nerfail_mlp_fwd)."""
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=pts.device)
    if getattr(fn, 'precision', 'f32') == 'f16x3':       # opt-in split-precision kernel (fp32-equivalent results)
        _lib.check(_lib.load().nerfail_mlp_fwd_f16(_lib.dev(fn.packed()), _lib.dev(fn.packed_f16()), fn.D, fn.W, fn._skip(),
                                                   _lib.dev(pts, 'pts'), _lib.dev(viewdirs, 'viewdirs'), R * N, N,
                                                   _lib.dev(raw), _lib.stream()))
        return raw
    _lib.check(_lib.load().nerfail_mlp_fwd(_lib.dev(fn.packed()), fn.D, fn.W, fn._skip(), _lib.dev(pts, 'pts'),
                                           _lib.dev(viewdirs, 'viewdirs'), R * N, N, _lib.dev(raw), _lib.stream()))
    return raw


def _mlp_rays(fn, rays, z_vals, acts=None):
    """Fused sample-point formation + encode + MLP: packed rays [R,11], z_vals [R,N] -> raw [R,N,4] (nerfail_mlp_fwd_rays: the
    points o + d z are formed inside the kernel and never touch HBM). acts: the training forward's activation buffer."""
    R, N = z_vals.shape
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=z_vals.device)
    _lib.check(_lib.load().nerfail_mlp_fwd_rays(_lib.dev(fn.packed()), fn.D, fn.W, fn._skip(), _lib.dev(rays, 'rays'),
                                                _lib.dev(z_vals, 'z_vals'), R, N, _lib.dev(raw), _lib.dev(acts), _lib.stream()))
    return raw


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """RN:37-51. With the stock encoders (multires 10 / 4) the encoding and the MLP are one kernel and
    the embedded tensor is never materialised; other encoders go through embed + NeRF.forward."""
    dev = _cuda()
    inputs = _lib.f32c(inputs, dev)
    if viewdirs is not None and _is_fused(embed_fn, embeddirs_fn, fn) and inputs.dim() == 3:
        return _mlp_points(fn, inputs, _lib.f32c(viewdirs, dev))
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


class FusedNetworkQuery:
    """The closure create_nerf builds at RN:201-204, as an object render_rays can recognise and bypass."""

    def __init__(self, embed_fn, embeddirs_fn, netchunk=1024 * 64):
        self.embed_fn, self.embeddirs_fn, self.netchunk = embed_fn, embeddirs_fn, netchunk

    def __call__(self, inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=self.embed_fn, embeddirs_fn=self.embeddirs_fn,
                           netchunk=self.netchunk)


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """RN:54-66."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (torch.cat(v, 0) if len(v) > 1 else v[0]) for k, v in all_ret.items()}


def _pack_rays(rays_o, rays_d, near, far):
    dev = _cuda()
    o = _lib.f32c(rays_o, dev).reshape(-1, 3)
    d = _lib.f32c(rays_d, dev).reshape(-1, 3)
    rays = torch.empty((o.shape[0], _lib.RAY_FLOATS), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().nerfail_pack_rays(_lib.dev(o), _lib.dev(d), o.shape[0], float(near), float(far),
                                             _lib.dev(rays), _lib.stream()))
    return rays


def ray_gen(H, W, K, c2w, near, far, pix_begin=0, pix_count=None):
    """K1 fused: packed rays [n, 11] of the pixel range [pix_begin, pix_begin + pix_count) of a view."""
    dev = _cuda()
    n = H * W - pix_begin if pix_count is None else pix_count
    rays = torch.empty((n, _lib.RAY_FLOATS), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().nerfail_ray_gen(int(H), int(W), _k4(K), _c2w12(c2w), float(near), float(far),
                                           int(pix_begin), int(n), _lib.dev(rays), _lib.stream()))
    return rays


def _render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, want_pts_max, kwargs):
    if ndc:
        raise NotImplementedError('ndc=True is LLFF-only (RN:112-114); the blender configs pass ndc=False')
    if not use_viewdirs:
        raise NotImplementedError('HIP path implements use_viewdirs=True (all configs/*.txt)')
    if c2w is not None:
        if c2w_staticcam is not None:                  # RN:105-107: rays from the static camera, dirs from c2w
            rays_flat = ray_gen(H, W, K, c2w_staticcam, near, far)
            rays_flat[:, 8:11] = ray_gen(H, W, K, c2w, near, far)[:, 8:11]
        else:
            rays_flat = ray_gen(H, W, K, c2w, near, far)
        sh = (H, W, 3)
    else:
        rays_o, rays_d = rays
        sh = tuple(rays_d.shape)
        rays_flat = _pack_rays(rays_o, rays_d, near, far)
    all_ret = batchify_rays(rays_flat, chunk, want_pts_max=want_pts_max, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map'] + (['pts_max'] if want_pts_max else [])
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """RN:69-134 -> [rgb_map, disp_map, acc_map, extras]."""
    return _render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, False, kwargs)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """RN:137-175 (driver loop; PNG writing needs imageio, which is optional)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    t = time.time()
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if savedir is not None:
            import imageio
            imageio.imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
    return np.stack(rgbs, 0), np.stack(disps, 0)


def create_nerf(args):
    """RN:178-259: builds coarse/fine NeRF, the query function, Adam, reloads the newest checkpoint."""
    dev = _cuda()
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
        grad_vars += list(model_fine.parameters())
    network_query_fn = FusedNetworkQuery(embed_fn, embeddirs_fn, args.netchunk)
    optimizer = Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))      # RN:207; one fused kernel per step
    start = 0
    basedir, expname = args.basedir, args.expname
    if getattr(args, 'ft_path', None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], map_location=dev)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


# ----------------------------------------------------------------------------- compositing
def _composite(raw, z_vals, rays, noise, white_bkgd, pts=None, want_pts_max=None):
    """pts_max (NC:418-423) is produced when want_pts_max (default: when pts is given); with pts None the kernel forms the
    point of the largest weight from the ray and its depth."""
    R, N = z_vals.shape
    dev = raw.device
    if want_pts_max is None:
        want_pts_max = pts is not None
    rgb_map = torch.empty((R, 3), dtype=torch.float32, device=dev)
    disp_map = torch.empty((R,), dtype=torch.float32, device=dev)
    acc_map = torch.empty((R,), dtype=torch.float32, device=dev)
    weights = torch.empty((R, N), dtype=torch.float32, device=dev)
    depth_map = torch.empty((R,), dtype=torch.float32, device=dev)
    pts_max = torch.empty((R, 3), dtype=torch.float32, device=dev) if want_pts_max else None
    _lib.check(_lib.load().nerfail_composite(
        _lib.dev(raw, 'raw'), _lib.dev(z_vals, 'z_vals'), _lib.dev(rays, 'rays'), _lib.dev(noise, 'noise'), R, N,
        int(bool(white_bkgd)), _lib.dev(rgb_map), _lib.dev(disp_map), _lib.dev(acc_map), _lib.dev(weights),
        _lib.dev(depth_map), _lib.dev(pts, 'pts') if want_pts_max else None, _lib.dev(pts_max), _lib.stream()))
    return rgb_map, disp_map, acc_map, weights, depth_map, pts_max


def _noise(shape, raw_noise_std, pytest, noise, dev):
    if noise is not None:
        return _lib.f32c(noise, dev).reshape(shape)
    if raw_noise_std > 0.:
        if pytest:                                     # RN:288-291 (np.random.rand, as the reference)
            np.random.seed(0)
            return torch.Tensor(np.random.rand(*shape) * raw_noise_std).to(dev).contiguous()
        return (torch.randn(shape, device=dev) * raw_noise_std).contiguous()
    return None


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None):
    """RN:262-305 -> (rgb_map, disp_map, acc_map, weights, depth_map)."""
    dev = _cuda()
    raw_in = raw
    raw, z_vals, rays_d = _lib.f32c(raw, dev), _lib.f32c(z_vals, dev), _lib.f32c(rays_d, dev)
    R = z_vals.shape[0]
    rays = torch.zeros((R, _lib.RAY_FLOATS), dtype=torch.float32, device=dev)
    rays[:, 3:6] = rays_d
    nz = _noise(tuple(z_vals.shape), raw_noise_std, pytest, noise, dev)
    if nz is None:
        # the registered op (torch.ops.nerfail_mi.composite): differentiable w.r.t. `raw` like the reference's function
        from . import ops  # noqa: F401
        raw_g = raw_in.to(dev).float().contiguous() if isinstance(raw_in, torch.Tensor) and raw_in.requires_grad else raw
        return tuple(torch.ops.nerfail_mi.composite(raw_g, z_vals, rays, bool(white_bkgd)))
    return _composite(raw, z_vals, rays, nz, white_bkgd)[:5]


# ----------------------------------------------------------------------------- the per-chunk pipeline
def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                want_pts_max=False, t_rand=None, u=None, noise=None, noise_fine=None):
    """RN:308-418 (and NC:320-436 when want_pts_max): coarse samples -> MLP -> composite ->
    importance samples (sorted merge) -> fine MLP -> composite (+ argmax-weight point).

    Extra keyword-only inputs beyond the reference: t_rand [R,N_samples], u [R,N_importance],
    noise / noise_fine [R,N] (already scaled) supply the random draws explicitly.
    """
    dev = _cuda()
    rays = _lib.f32c(ray_batch, dev)
    if rays.shape[-1] != _lib.RAY_FLOATS:
        raise NotImplementedError('ray_batch must be [R, 11] (use_viewdirs=True packing, RN:116-123)')
    R = rays.shape[0]
    q = network_query_fn
    fused = q is None or (isinstance(q, FusedNetworkQuery) and _is_fused(q.embed_fn, q.embeddirs_fn, network_fn))

    if perturb > 0. and t_rand is None:
        if pytest:                                     # RN:374-377
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(R, N_samples))
        else:
            t_rand = torch.rand((R, N_samples), device=dev)
    t_rand = _lib.f32c(t_rand, dev) if perturb > 0. else None
    if N_importance > 0 and u is None:
        det = (perturb == 0.)
        if pytest:                                     # RH:215-223
            np.random.seed(0)
            u = torch.Tensor(np.linspace(0., 1., N_importance)) if det else torch.Tensor(np.random.rand(R, N_importance))
        else:
            u = linspace01(N_importance, dev) if det else torch.rand((R, N_importance), device=dev)
    if u is not None:
        u = _lib.f32c(u, dev)
    nz = _noise((R, N_samples), raw_noise_std, pytest, noise, dev)
    nzf = _noise((R, N_samples + N_importance), raw_noise_std, pytest, noise_fine, dev) if N_importance > 0 else None

    def pipeline(rays, train=False):
        return _render_rays_pipeline(rays, network_fn, network_fine, q if not fused else None, N_samples, N_importance,
                                     lindisp, white_bkgd, t_rand, u, nz, nzf, want_pts_max, train)

    needs_grad = fused and torch.is_grad_enabled() and any(
        p.requires_grad for n in (network_fn, network_fine) if n is not None for p in n.parameters())
    if needs_grad:
        from . import _train
        nets = [network_fn] + ([network_fine] if network_fine is not None else [])
        params = [p for n in nets for p in _train.ordered_params(n)]
        cfg = dict(pipeline=pipeline, network_fn=network_fn, network_fine=network_fine, white_bkgd=white_bkgd,
                   retraw=retraw)
        o = _train.RenderRaysTrain.apply(rays, cfg, *params)
        out = dict(zip(('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'pts_max', 'raw'), o))
    else:
        out = pipeline(rays, False)

    ret = {'rgb_map': out['rgb_map'], 'disp_map': out['disp_map'], 'acc_map': out['acc_map']}
    if want_pts_max:
        ret['pts_max'] = out['pts_max']
    if retraw:
        ret['raw'] = out['raw']
    if N_importance > 0:
        ret.update(rgb0=out['rgb0'], disp0=out['disp0'], acc0=out['acc0'], z_std=out['z_std'])
    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def _render_rays_pipeline(rays, network_fn, network_fine, user_query, N_samples, N_importance, lindisp, white_bkgd,
                          t_rand, u, nz, nzf, want_pts_max, train):
    """The kernel sequence of one chunk. `train` saves what the backward needs (activations, raw, z)."""
    dev = rays.device
    lib = _lib.load()
    R = rays.shape[0]
    st = _lib.stream()
    viewdirs = rays[:, 8:11].contiguous()

    def empty():
        return torch.empty((0,), dtype=torch.float32, device=dev)

    # The fused exact-f32 path never materialises the sample points: the MLP kernel forms pts = o + d z from the ray and the
    # depth it already addresses (BASELINE north_star: "fused sample+encode+MLP"). A user-supplied query function and the
    # opt-in split-precision kernels still receive a [R,N,3] point tensor.
    def wants_pts(fn):
        return user_query is not None or getattr(fn, 'precision', 'f32') == 'f16x3'

    def query(p, z, fn):
        if user_query is not None:
            return _lib.f32c(user_query(p, viewdirs, fn), dev), None   # user-supplied query function (reference contract)
        if train:
            from . import _train
            a_ = acts_for.pop(0) if acts_for else None
            if p is None:
                return _train.mlp_fwd_train_rays(fn, rays, z, acts=a_)
            return _train.mlp_fwd_train(fn, p, viewdirs, acts=a_)
        if p is None:
            return _mlp_rays(fn, rays, z), None                        # points formed in the kernel
        return _mlp_points(fn, p, viewdirs), None                      # fused encode + MLP, nothing materialised

    # training: the saved activations of the coarse and the fine pass share ONE buffer (coarse tiles first), so that the
    # backward walks both networks in one launch per kernel
    acts_all, acts_for = None, []
    if train and user_query is None and N_importance > 0:
        from . import _train
        run_fn_ = network_fn if network_fine is None else network_fine
        if _train._same_arch(network_fn, run_fn_) and (R * N_samples) % 32 == 0:
            nc, nf = _train.acts_floats(network_fn, R * N_samples), _train.acts_floats(run_fn_, R * (N_samples + N_importance))
            acts_all = torch.empty((nc + nf,), dtype=torch.float32, device=dev)
            acts_for = [acts_all[:nc], acts_all[nc:]]

    z_vals = torch.empty((R, N_samples), dtype=torch.float32, device=dev)
    pts = torch.empty((R, N_samples, 3), dtype=torch.float32, device=dev) if wants_pts(network_fn) else None
    t_lin = linspace01(N_samples, dev)      # (cached per device; bound to a name so that the pointer below is never a temporary's)
    _lib.check(lib.nerfail_sample_coarse(_lib.dev(rays), R, _lib.dev(t_lin), int(N_samples),
                                         _lib.dev(t_rand, 't_rand'), int(bool(lindisp)), _lib.dev(z_vals), _lib.dev(pts), st))
    raw, acts = query(pts, z_vals, network_fn)
    last_pass = not (N_importance > 0)
    rgb_map, disp_map, acc_map, weights, depth_map, pts_max = _composite(
        raw, z_vals, rays, nz, white_bkgd, pts, want_pts_max and last_pass)
    out = {'rgb0': empty(), 'disp0': empty(), 'acc0': empty(), 'z_std': empty()}
    saved = {'rays': rays, 'coarse': dict(raw=raw, z=z_vals, acts=acts, noise=nz), 'fine': None, 'acts_all': acts_all}
    if N_importance > 0:
        out.update(rgb0=rgb_map, disp0=disp_map, acc0=acc_map)
        Nt = N_samples + N_importance
        z_samples = torch.empty((R, N_importance), dtype=torch.float32, device=dev)
        z_fine = torch.empty((R, Nt), dtype=torch.float32, device=dev)
        run_fn = network_fn if network_fine is None else network_fine
        pts = torch.empty((R, Nt, 3), dtype=torch.float32, device=dev) if wants_pts(run_fn) else None
        z_std = torch.empty((R,), dtype=torch.float32, device=dev)
        _lib.check(lib.nerfail_sample_fine(_lib.dev(rays), R, _lib.dev(z_vals), _lib.dev(weights), int(N_samples),
                                           _lib.dev(u, 'u'), int(u.dim() == 1), int(N_importance), _lib.dev(z_samples),
                                           _lib.dev(z_fine), _lib.dev(pts), _lib.dev(z_std), st))
        z_vals = z_fine
        raw, acts = query(pts, z_vals, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map, pts_max = _composite(
            raw, z_vals, rays, nzf, white_bkgd, pts, want_pts_max)
        out['z_std'] = z_std
        saved['fine'] = dict(raw=raw, z=z_vals, acts=acts, noise=nzf)
    out.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map, raw=raw,
               pts_max=pts_max if pts_max is not None else empty())
    if train:
        out['_saved'] = saved
    return out
