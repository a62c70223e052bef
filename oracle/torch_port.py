"""ORACLE (test infrastructure, not product code): PyTorch-CPU port of the training step, fp32 + autograd.

Only tests/ and bench.py's cpu_baseline legs may import this module. It exists for the CPU BASELINE of the fwd+bwd
metric (BASELINE.md section 3 plans a pure-PyTorch CPU restatement timed with torch.set_num_threads): the numpy oracle
(oracle/nerf.py::train_step_grads) accumulates its gradients in float64 for accuracy and is ~5x slower than the
reference, which would flatter the GPU/CPU ratio. Same arithmetic path as the reference (nn.functional.linear, relu,
cumprod, searchsorted, sort; autograd): RH = Create_spatial_point_set/nerf_pytorch/run_nerf_helpers.py, RN = .../run_nerf.py.
Pinned by tests/test_oracle_nerf.py::test_torch_port_matches_reference_training_step (fixture g7).
"""
import torch
import torch.nn.functional as F


def embed(x, L):
    """RH:15-50: [x, sin(x 2^k), cos(x 2^k)]_k."""
    out = [x]
    for k in range(L):
        f = float(2 ** k)
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


def nerf_forward(p, pts, viewdirs, D=8, skips=(4,)):
    """RH:100-123 on [R,N,3] points; p = dict of parameter tensors with the reference's state_dict keys."""
    R, N = pts.shape[:2]
    e_p = embed(pts.reshape(-1, 3), 10)
    e_d = embed(viewdirs[:, None, :].expand(R, N, 3).reshape(-1, 3), 4)
    h = e_p
    for i in range(D):
        h = F.relu(F.linear(h, p['pts_linears.%d.weight' % i], p['pts_linears.%d.bias' % i]))
        if i in skips and i < D - 1:
            h = torch.cat([e_p, h], -1)
    alpha = F.linear(h, p['alpha_linear.weight'], p['alpha_linear.bias'])
    feature = F.linear(h, p['feature_linear.weight'], p['feature_linear.bias'])
    hv = F.relu(F.linear(torch.cat([feature, e_d], -1), p['views_linears.0.weight'], p['views_linears.0.bias']))
    rgb = F.linear(hv, p['rgb_linear.weight'], p['rgb_linear.bias'])
    return torch.cat([rgb, alpha], -1).reshape(R, N, 4)


def raw2outputs(raw, z_vals, rays_d, white_bkgd):
    """RN:262-305 (no noise)."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1. - torch.exp(-F.relu(raw[..., 3]) * dists)
    weights = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, weights


def sample_pdf(bins, weights, u):
    """RH:200-243 with explicit draws u [R,N]."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


def train_step(ray_batch, sd_coarse, sd_fine, target, t_rand, u, N_samples=64, N_importance=128, D=8, white_bkgd=True):
    """RN:776-791: loss = mse(rgb, target) + mse(rgb0, target), loss.backward(). Returns (loss, grads_coarse, grads_fine)
    as python float / dicts of numpy arrays keyed like the reference's state_dict."""
    T = torch.from_numpy
    pc = {k: T(v).clone().requires_grad_(True) for k, v in sd_coarse.items()}
    pf = {k: T(v).clone().requires_grad_(True) for k, v in sd_fine.items()}
    rays = T(ray_batch)
    rays_o, rays_d, viewdirs = rays[:, 0:3], rays[:, 3:6], rays[:, -3:]
    near, far = rays[:, 6:7], rays[:, 7:8]
    t_vals = torch.linspace(0., 1., steps=N_samples)
    z_vals = (near * (1. - t_vals) + far * t_vals).expand(rays.shape[0], N_samples)
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper, lower = torch.cat([mids, z_vals[..., -1:]], -1), torch.cat([z_vals[..., :1], mids], -1)
    z_vals = lower + (upper - lower) * T(t_rand)                                          # RN:367-379 (perturb = 1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    rgb0, w0 = raw2outputs(nerf_forward(pc, pts, viewdirs, D), z_vals, rays_d, white_bkgd)
    z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_samples = sample_pdf(z_mid, w0[..., 1:-1], T(u)).detach()                           # RN:392-394
    z_fine, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_fine[..., :, None]
    rgb1, _ = raw2outputs(nerf_forward(pf, pts, viewdirs, D), z_fine, rays_d, white_bkgd)
    tg = T(target)
    loss = torch.mean((rgb1 - tg) ** 2) + torch.mean((rgb0 - tg) ** 2)
    loss.backward()
    return (float(loss.detach()), {k: v.grad.numpy() for k, v in pc.items()}, {k: v.grad.numpy() for k, v in pf.items()})
