"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Everything here is numpy-only and platform independent (legacy MT19937 RandomState), so a
fixture can store just the seed instead of megabytes of weights. Shapes/values follow the
measurement plan of SURVEY.md section 8(d) (lego-like camera, near 2 / far 6, alpha bias +0.5).
"""
import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618


def lego_intrinsics(H=800, W=800):
    """focal = .5*W/tan(.5*camera_angle_x); K as run_nerf.py:632-636 builds it."""
    focal = .5 * W / np.tan(.5 * LEGO_CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    return focal, K


def pose_spherical(theta, phi, radius):
    """Same matrix product as load_blender.py:8-34 (float32), returns [4,4] float32."""
    t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    ph = phi / 180. * np.pi
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                   [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], np.float32)
    th = theta / 180. * np.pi
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                   [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return (flip @ (rt @ (rp @ t))).astype(np.float32)


def nerf_state_dict(D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), seed=0,
                    alpha_bias=0.5, gain=1.0):
    """Seeded NeRF weights with the reference's state_dict keys (run_nerf_helpers.py:83-98).

    nn.Linear-style uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) init from a numpy stream so that
    the generator, the tests and the GPU box all rebuild bit-identical weights from `seed`.
    `alpha_linear.bias` is raised so the density is non-trivial.
    """
    rs = np.random.RandomState(seed)
    sd = {}

    def lin(name, fan_in, fan_out):
        b = gain / np.sqrt(fan_in)
        sd[name + '.weight'] = rs.uniform(-b, b, size=(fan_out, fan_in)).astype(np.float32)
        sd[name + '.bias'] = rs.uniform(-b, b, size=(fan_out,)).astype(np.float32)

    lin('pts_linears.0', input_ch, W)
    for i in range(D - 1):
        lin('pts_linears.%d' % (i + 1), W + input_ch if i in skips else W, W)
    lin('views_linears.0', input_ch_views + W, W // 2)
    lin('feature_linear', W, W)
    lin('alpha_linear', W, 1)
    lin('rgb_linear', W // 2, 3)
    sd['alpha_linear.bias'] = (sd['alpha_linear.bias'] + np.float32(alpha_bias)).astype(np.float32)
    return sd


def ray_batch(n_rays, seed=0, H=800, W=800, pose_theta=-180.0, near=2.0, far=6.0):
    """[n_rays, 11] float32 rays of a lego-like view: o(3) d(3) near far viewdir(3).

    Pixels are drawn at random from the central half of the image so most rays hit the
    unit-ish volume the seeded MLP has density in.
    """
    rs = np.random.RandomState(seed)
    focal, K = lego_intrinsics(H, W)
    c2w = pose_spherical(pose_theta, -30.0, 4.0)[:3, :4]
    jj = rs.randint(H // 4, 3 * H // 4, size=n_rays)
    ii = rs.randint(W // 4, 3 * W // 4, size=n_rays)
    dirs = np.stack([(ii.astype(np.float32) - np.float32(K[0][2])) / np.float32(K[0][0]),
                     -(jj.astype(np.float32) - np.float32(K[1][2])) / np.float32(K[1][1]),
                     -np.ones(n_rays, np.float32)], -1).astype(np.float32)
    rays_d = (dirs[:, None, :] * c2w[:3, :3][None]).sum(-1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    vd = (rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True)).astype(np.float32)
    nf = np.tile(np.array([[near, far]], np.float32), (n_rays, 1))
    return np.concatenate([rays_o, rays_d, nf, vd], -1).astype(np.float32)


def sphere_shell_points(n, seed=0, radius=1.0, jitter=0.01):
    """Synthetic pts_max-like point set: shell of radius 1 +- jitter."""
    rs = np.random.RandomState(seed)
    v = rs.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = radius + jitter * (2 * rs.uniform(size=(n, 1)) - 1)
    return (v * r).astype(np.float32)


def disc_alpha_image(B, H, W, seed=0, radius_frac=0.375):
    """uint8-valued float32 BGRA images, alpha=255 inside a centred disc (radius 300 px at 800)."""
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, size=(B, H, W, 4)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    inside = ((yy - H / 2 + .5) ** 2 + (xx - W / 2 + .5) ** 2) <= (radius_frac * H) ** 2
    img[..., 3] = np.where(inside, 255.0, 0.0)[None]
    return img


def sphere_view_points(H, W, theta, phi=-30.0, radius=4.0, obj_radius=1.0, rough=0.01, near=2.0):
    """Analytic `pts_max` [H,W,3] of a rough unit sphere seen from pose_spherical(theta, phi, radius) (NC:418-423
    semantics): a pixel whose ray hits the sphere gets its first intersection point (surface radius 1 +- rough, a
    deterministic per-pixel hash), a pixel that misses gets the near-plane point o + near*d (argmax of all-zero weights is
    sample 0). Unlike sphere_shell_points the ARRAY ORDER is the pixel order, as in the real pipeline: neighbouring
    pixels are neighbouring 3-D points, so an 8-NN map built on these has the index locality of a real scene."""
    focal, K = lego_intrinsics(H, W)
    c2w = pose_spherical(theta, phi, radius)[:3, :4].astype(np.float64)
    jj, ii = np.mgrid[0:H, 0:W]
    dirs = np.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -np.ones_like(ii, dtype=np.float64)], -1)
    d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    o = c2w[:3, 3]
    pix = (jj * W + ii).astype(np.uint64)
    hsh = ((pix * np.uint64(2654435761) + np.uint64(int(abs(theta) * 1000) + 12345)) % np.uint64(1 << 20)).astype(np.float64) / (1 << 20)
    r = obj_radius * (1.0 + rough * (2.0 * hsh - 1.0))
    a = (d * d).sum(-1)
    b = 2.0 * (d * o).sum(-1)
    c = (o * o).sum() - r * r
    disc = b * b - 4 * a * c
    hit = disc > 0
    t = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), near)
    return (o + d * t[..., None]).astype(np.float32)
