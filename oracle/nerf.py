"""ORACLE (test infrastructure, not product code): CPU/numpy restatement of the NeRF render path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It is the checker for the HIP kernels, never the thing shipped or measured as `value`.

Pinned: every function here is checked against golden vectors produced by RUNNING the
reference on CPU in the build container (tests/golden/make_golden.py, fixtures g1..g7).

All arithmetic is float32, in the reference's operation order where that order is defined
(no fused multiply-add: numpy has none). Citations are relative to /root/reference:
RN = Create_spatial_point_set/nerf_pytorch/run_nerf.py, RH = .../run_nerf_helpers.py,
NC = Create_spatial_point_set/nerf_to_coord.py.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- rays
def get_rays(H, W, K, c2w):
    """RH:157-166. i = column, j = row; dirs = ((i-cx)/fx, -(j-cy)/fy, -1); rays_d = R @ dirs."""
    c2w = np.asarray(c2w, F32)
    i, j = np.meshgrid(np.arange(W, dtype=F32), np.arange(H, dtype=F32), indexing='xy')
    # K is float64 numpy in the reference; torch demotes the scalar to the tensor dtype (fp32)
    cx, cy, fx, fy = F32(K[0][2]), F32(K[1][2]), F32(K[0][0]), F32(K[1][1])
    dirs = np.stack([(i - cx) / fx, -(j - cy) / fy, -np.ones_like(i)], -1)
    prod = dirs[..., None, :] * c2w[:3, :3]                      # [H,W,3,3]
    rays_d = (prod[..., 0] + prod[..., 1]) + prod[..., 2]       # torch.sum over 3 elements
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).copy()
    return rays_o.astype(F32), rays_d.astype(F32)


def pack_rays(rays_o, rays_d, near, far):
    """RN:102-123 with use_viewdirs=True, ndc=False: [R,11] = o d near far viewdir."""
    rays_o = np.asarray(rays_o, F32).reshape(-1, 3)
    rays_d = np.asarray(rays_d, F32).reshape(-1, 3)
    nrm = np.sqrt((rays_d * rays_d).sum(-1, keepdims=True, dtype=F32)).astype(F32)
    viewdirs = (rays_d / nrm).astype(F32)
    ones = np.ones_like(rays_d[..., :1])
    return np.concatenate([rays_o, rays_d, F32(near) * ones, F32(far) * ones, viewdirs], -1).astype(F32)


# ----------------------------------------------------------------------------- positional encoding
def embed(x, L):
    """RH:15-50: [x, sin(x*2^0), cos(x*2^0), ..., sin(x*2^(L-1)), cos(x*2^(L-1))] -> 3+6L channels."""
    x = np.asarray(x, F32)
    outs = [x]
    for k in range(L):
        f = F32(2.0 ** k)
        xf = (x * f).astype(F32)
        outs.append(np.sin(xf).astype(F32))
        outs.append(np.cos(xf).astype(F32))
    return np.concatenate(outs, -1).astype(F32)


# ----------------------------------------------------------------------------- MLP
def nerf_forward(sd, emb, D=8, W=256, skips=(4,), input_ch=63, return_acts=False):
    """RH:100-123 with use_viewdirs=True. `sd` = state_dict as numpy (keys RH:83-98).

    emb [M, 63+27] -> raw [M,4] = [rgb(3), alpha(1)]."""
    emb = np.asarray(emb, F32)
    input_pts, input_views = emb[:, :input_ch], emb[:, input_ch:]
    h = input_pts
    acts = []
    for i in range(D):
        h = h @ sd['pts_linears.%d.weight' % i].T + sd['pts_linears.%d.bias' % i]
        h = np.maximum(h, F32(0))
        if i in skips:
            h = np.concatenate([input_pts, h], -1)
        acts.append(h)
    alpha = h @ sd['alpha_linear.weight'].T + sd['alpha_linear.bias']
    feature = h @ sd['feature_linear.weight'].T + sd['feature_linear.bias']
    hv = np.concatenate([feature, input_views], -1)
    hv = np.maximum(hv @ sd['views_linears.0.weight'].T + sd['views_linears.0.bias'], F32(0))
    rgb = hv @ sd['rgb_linear.weight'].T + sd['rgb_linear.bias']
    out = np.concatenate([rgb, alpha], -1).astype(F32)
    if return_acts:
        return out, acts, feature, hv
    return out


def run_network(sd, pts, viewdirs, D=8, W=256, skips=(4,), netchunk=1 << 16):
    """RN:37-51: flatten pts [R,N,3], embed pts (L=10) and per-sample-expanded viewdirs (L=4)."""
    pts = np.asarray(pts, F32)
    R, N = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    dirs = np.broadcast_to(np.asarray(viewdirs, F32)[:, None, :], pts.shape).reshape(-1, 3)
    out = np.empty((flat.shape[0], 4), F32)
    for s in range(0, flat.shape[0], netchunk):                  # batchify RN:27-34
        e = np.concatenate([embed(flat[s:s + netchunk], 10), embed(dirs[s:s + netchunk], 4)], -1)
        out[s:s + netchunk] = nerf_forward(sd, e, D=D, W=W, skips=skips)
    return out.reshape(R, N, 4)


# ----------------------------------------------------------------------------- compositing
def raw2outputs(raw, z_vals, rays_d, noise=None, white_bkgd=False):
    """RN:262-305. `noise` is the already scaled term added to raw[...,3] (randn*raw_noise_std)."""
    raw = np.asarray(raw, F32)
    z_vals = np.asarray(z_vals, F32)
    rays_d = np.asarray(rays_d, F32)
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = np.concatenate([dists, np.full_like(dists[..., :1], 1e10)], -1)
    nrm = np.sqrt((rays_d * rays_d).sum(-1, dtype=F32)).astype(F32)
    dists = (dists * nrm[..., None]).astype(F32)
    with np.errstate(over='ignore'):
        rgb = (F32(1) / (F32(1) + np.exp(-raw[..., :3]))).astype(F32)
    sigma = raw[..., 3] if noise is None else (raw[..., 3] + np.asarray(noise, F32))
    alpha = (F32(1) - np.exp(-np.maximum(sigma, F32(0)) * dists)).astype(F32)
    t = np.concatenate([np.ones_like(alpha[..., :1]), (F32(1) - alpha) + F32(1e-10)], -1)
    # torch's CPU cumprod accumulates in double (at::acc_type<float, false>) and rounds per element
    trans = np.cumprod(t.astype(np.float64), -1).astype(F32)[..., :-1]
    weights = (alpha * trans).astype(F32)
    rgb_map = (weights[..., None] * rgb).sum(-2, dtype=F32)
    depth_map = (weights * z_vals).sum(-1, dtype=F32)
    acc_map = weights.sum(-1, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = depth_map / acc_map
        # torch.max propagates NaN (0/0 when acc == 0): RN:299
        disp_map = F32(1) / np.where(np.isnan(ratio), ratio, np.maximum(F32(1e-10), ratio))
    if white_bkgd:
        rgb_map = rgb_map + (F32(1) - acc_map[..., None])
    return rgb_map.astype(F32), disp_map.astype(F32), acc_map.astype(F32), weights, depth_map.astype(F32)


# ----------------------------------------------------------------------------- hierarchical sampling
def sample_pdf(bins, weights, N_samples, u=None):
    """RH:200-243. u=None -> det (linspace(0,1,N)); else explicit uniform draws [R,N_samples]."""
    bins = np.asarray(bins, F32)
    weights = np.asarray(weights, F32) + F32(1e-5)
    pdf = (weights / weights.sum(-1, keepdims=True, dtype=F32)).astype(F32)
    # torch's CPU cumsum accumulates in double (at::acc_type<float, false>) and rounds per element
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(F32)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = np.broadcast_to(torch_linspace01(N_samples), cdf.shape[:-1] + (N_samples,))
    u = np.ascontiguousarray(u, F32)
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)            # searchsorted(right=True)
    below = np.maximum(0, inds - 1)
    above = np.minimum(cdf.shape[-1] - 1, inds)
    cdf_b = np.take_along_axis(cdf, below, -1)
    cdf_a = np.take_along_axis(cdf, above, -1)
    bin_b = np.take_along_axis(bins, below, -1)
    bin_a = np.take_along_axis(bins, above, -1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < F32(1e-5), F32(1), denom)
    t = (u - cdf_b) / denom
    return (bin_b + t * (bin_a - bin_b)).astype(F32)


def torch_linspace01(steps):
    """torch.linspace(0., 1., steps) in float32: the CPU kernel uses a float32 step and a fused
    (single rounding) start + step*i, symmetric halves around the middle (RangeFactories.cpp)."""
    step = np.float64(F32(1.0) / F32(steps - 1))
    i = np.arange(steps)
    return np.where(i < steps // 2, 0.0 + step * i, 1.0 - step * (steps - 1 - i)).astype(F32)


def coarse_z_vals(near, far, N_samples, t_rand=None, lindisp=False):
    """RN:357-379. near/far [R,1]; t_rand [R,N] explicit stratified draws or None."""
    t_vals = torch_linspace01(N_samples)
    near, far = np.asarray(near, F32), np.asarray(far, F32)
    if not lindisp:
        z = near * (F32(1) - t_vals) + far * t_vals
    else:
        z = F32(1) / (F32(1) / near * (F32(1) - t_vals) + F32(1) / far * t_vals)
    z = np.broadcast_to(z, (near.shape[0], N_samples)).astype(F32)
    if t_rand is not None:
        mids = F32(.5) * (z[..., 1:] + z[..., :-1])
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = (lower + (upper - lower) * np.asarray(t_rand, F32)).astype(F32)
    return z


def render_rays(ray_batch, sd_coarse, N_samples, N_importance=0, sd_fine=None, white_bkgd=False,
                t_rand=None, u=None, D=8, W=256, lindisp=False, want_pts_max=True, return_weights=False,
                noise=None, noise_fine=None):
    """RN:308-418 (+ NC:418-423 pts_max). perturb>0 <=> t_rand given; det sampling <=> u is None;
    noise / noise_fine = the (already scaled) raw_noise_std draws of RN:285 for the coarse / fine composite."""
    ray_batch = np.asarray(ray_batch, F32)
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z_vals = coarse_z_vals(near, far, N_samples, t_rand, lindisp)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).astype(F32)
    raw = run_network(sd_coarse, pts, viewdirs, D=D, W=W)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, noise, white_bkgd)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb_map, disp0=disp_map, acc0=acc_map)
        if return_weights:
            ret['weights0'], ret['z_vals0'], ret['raw0'] = weights, z_vals, raw
        z_mid = F32(.5) * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, u=u)
        z_vals = np.sort(np.concatenate([z_vals, z_samples], -1), -1)
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).astype(F32)
        raw = run_network(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs, D=D, W=W)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, noise_fine, white_bkgd)
        zs = z_samples.astype(F32)
        mean = zs.mean(-1, keepdims=True, dtype=F32)
        ret['z_std'] = np.sqrt(((zs - mean) ** 2).mean(-1, dtype=F32)).astype(F32)   # RN:412
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map, raw=raw)
    if return_weights:
        ret['weights'], ret['z_vals'] = weights, z_vals
    if want_pts_max:                                                                 # NC:418-423
        am = np.argmax(weights, axis=1)
        ret['pts_max'] = pts[np.arange(pts.shape[0]), am]
    return ret


def render(H, W, K, c2w, near, far, sd_coarse, sd_fine, N_samples=64, N_importance=128,
           white_bkgd=True, chunk=1 << 15, D=8, Wn=256, rays_slice=None):
    """RN:69-134 / NC:70-135 for the blender configs (use_viewdirs, ndc=False)."""
    rays_o, rays_d = get_rays(H, W, K, c2w)
    rays = pack_rays(rays_o, rays_d, near, far)
    if rays_slice is not None:
        rays = rays[rays_slice]
    outs = []
    for s in range(0, rays.shape[0], chunk):                                         # RN:54-66
        outs.append(render_rays(rays[s:s + chunk], sd_coarse, N_samples, N_importance, sd_fine,
                                white_bkgd, D=D, W=Wn))
    return {k: np.concatenate([o[k] for o in outs], 0) for k in outs[0]}


# ----------------------------------------------------------------------------- training step (fwd + bwd)
def _mlp_forward_cached(sd, pts, viewdirs, D, W, skips=(4,)):
    R, N = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    dirs = np.broadcast_to(np.asarray(viewdirs, F32)[:, None, :], pts.shape).reshape(-1, 3)
    e_p, e_d = embed(flat, 10), embed(dirs, 4)
    ins, pre = [], []
    h = e_p
    for i in range(D):
        ins.append(h)
        z = h @ sd['pts_linears.%d.weight' % i].T + sd['pts_linears.%d.bias' % i]
        pre.append(z)
        h = np.maximum(z, F32(0))
        if i in skips and i < D - 1:
            h = np.concatenate([e_p, h], -1)
    alpha = h @ sd['alpha_linear.weight'].T + sd['alpha_linear.bias']
    feature = h @ sd['feature_linear.weight'].T + sd['feature_linear.bias']
    hv_in = np.concatenate([feature, e_d], -1)
    hv_pre = hv_in @ sd['views_linears.0.weight'].T + sd['views_linears.0.bias']
    hv = np.maximum(hv_pre, F32(0))
    rgb = hv @ sd['rgb_linear.weight'].T + sd['rgb_linear.bias']
    raw = np.concatenate([rgb, alpha], -1).astype(F32)
    return raw.reshape(R, N, 4), dict(ins=ins, pre=pre, h_last=h, hv_in=hv_in, hv_pre=hv_pre, hv=hv)


def mlp_backward(sd, cache, d_raw, D, W, skips=(4,)):
    """Gradients of all NeRF parameters given d_raw [M,4] (autograd of RH:100-123); float64 accumulation."""
    f8 = np.float64
    d_raw = np.asarray(d_raw, f8).reshape(-1, 4)
    g = {}
    d_rgb, d_alpha = d_raw[:, :3], d_raw[:, 3:4]
    g['rgb_linear.weight'] = d_rgb.T @ cache['hv'].astype(f8)
    g['rgb_linear.bias'] = d_rgb.sum(0)
    d_hv = (d_rgb @ sd['rgb_linear.weight'].astype(f8)) * (cache['hv_pre'] > 0)
    g['views_linears.0.weight'] = d_hv.T @ cache['hv_in'].astype(f8)
    g['views_linears.0.bias'] = d_hv.sum(0)
    d_feat = (d_hv @ sd['views_linears.0.weight'].astype(f8))[:, :W]
    h = cache['h_last'].astype(f8)
    g['feature_linear.weight'] = d_feat.T @ h
    g['feature_linear.bias'] = d_feat.sum(0)
    g['alpha_linear.weight'] = d_alpha.T @ h
    g['alpha_linear.bias'] = d_alpha.sum(0)
    d_h = d_feat @ sd['feature_linear.weight'].astype(f8) + d_alpha @ sd['alpha_linear.weight'].astype(f8)
    for i in reversed(range(D)):
        if i in skips and i < D - 1:
            d_h = d_h[:, 63:]                     # the concatenated input_pts part carries no parameters
        d_z = d_h * (cache['pre'][i] > 0)
        g['pts_linears.%d.weight' % i] = d_z.T @ cache['ins'][i].astype(f8)
        g['pts_linears.%d.bias' % i] = d_z.sum(0)
        d_h = d_z @ sd['pts_linears.%d.weight' % i].astype(f8)
    return {k: v.astype(F32) for k, v in g.items()}


def raw2outputs_backward(raw, z_vals, rays_d, d_rgb_map, white_bkgd=True):
    """d(loss)/d(raw) for a loss that depends on rgb_map only (RN:781-789); autograd of RN:262-305."""
    f8 = np.float64
    raw = np.asarray(raw, F32)
    z_vals = np.asarray(z_vals, F32)
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = np.concatenate([dists, np.full_like(dists[..., :1], 1e10)], -1)
    nrm = np.sqrt((np.asarray(rays_d, F32) ** 2).sum(-1, dtype=F32)).astype(F32)
    dists = (dists * nrm[..., None]).astype(f8)
    c = 1.0 / (1.0 + np.exp(-raw[..., :3].astype(f8)))
    sig = raw[..., 3].astype(f8)
    with np.errstate(over='ignore'):
        e = np.exp(-np.maximum(sig, 0.0) * dists)
    alpha = (F32(1) - e.astype(F32)).astype(f8)                       # forward values are float32
    t = (1.0 - alpha) + 1e-10
    T = np.concatenate([np.ones_like(alpha[..., :1]), np.cumprod(t, -1)[..., :-1]], -1)
    w = alpha * T
    g = np.asarray(d_rgb_map, f8)[:, None, :]                         # [R,1,3]
    gw = (g * c).sum(-1) - (g.sum(-1) if white_bkgd else 0.0)        # dL/dw_i
    gwW = gw * w
    S = np.flip(np.cumsum(np.flip(gwW, -1), -1), -1) - gwW            # sum_{k>i} gw_k w_k
    d_alpha = gw * T - S / t
    d_sigma = d_alpha * dists * (1.0 - alpha) * (sig > 0)
    d_rgbraw = (w[..., None] * g) * c * (1.0 - c)
    return np.concatenate([d_rgbraw, d_sigma[..., None]], -1).astype(F32)


def train_step_grads(ray_batch, sd_coarse, sd_fine, target, N_samples=64, N_importance=128, t_rand=None, u=None,
                     D=8, W=256, white_bkgd=True):
    """One training step's loss and parameter gradients (RN:776-791): loss = mse(rgb, t) + mse(rgb0, t)."""
    ray_batch = np.asarray(ray_batch, F32)
    rays_o, rays_d, viewdirs = ray_batch[:, 0:3], ray_batch[:, 3:6], ray_batch[:, -3:]
    z0 = coarse_z_vals(ray_batch[:, 6:7], ray_batch[:, 7:8], N_samples, t_rand)
    pts0 = (rays_o[:, None, :] + rays_d[:, None, :] * z0[:, :, None]).astype(F32)
    raw0, cache0 = _mlp_forward_cached(sd_coarse, pts0, viewdirs, D, W)
    rgb0, _, _, w0, _ = raw2outputs(raw0, z0, rays_d, None, white_bkgd)
    z_mid = F32(.5) * (z0[..., 1:] + z0[..., :-1])
    z_s = sample_pdf(z_mid, w0[..., 1:-1], N_importance, u=u)
    z1 = np.sort(np.concatenate([z0, z_s], -1), -1)
    pts1 = (rays_o[:, None, :] + rays_d[:, None, :] * z1[:, :, None]).astype(F32)
    raw1, cache1 = _mlp_forward_cached(sd_fine, pts1, viewdirs, D, W)
    rgb1, _, _, _, _ = raw2outputs(raw1, z1, rays_d, None, white_bkgd)
    target = np.asarray(target, F32)
    loss = float(np.mean((rgb1 - target) ** 2, dtype=np.float64) + np.mean((rgb0 - target) ** 2, dtype=np.float64))
    n = rgb1.size
    d1 = raw2outputs_backward(raw1, z1, rays_d, 2.0 * (rgb1 - target) / n, white_bkgd)
    d0 = raw2outputs_backward(raw0, z0, rays_d, 2.0 * (rgb0 - target) / n, white_bkgd)
    return dict(loss=loss, rgb_map=rgb1, rgb0=rgb0, d_raw_fine=d1, d_raw_coarse=d0,
                grads_fine=mlp_backward(sd_fine, cache1, d1, D, W), grads_coarse=mlp_backward(sd_coarse, cache0, d0, D, W))
