"""`torch.ops.nerfail_mi.*`: the hot-path kernels registered with the PyTorch dispatcher (SURVEY.md section 8b "What the
C-ABI layer must export"; BASELINE north_star "exposed to Python through PyTorch-ROCm custom ops").

Each op is a `torch.library.custom_op` over ONE entry point of libnerfail_hip.so (include/nerfail_hip.h), with
  * a device implementation for 'cuda' (= HIP on ROCm): dense float32 HIP tensors in, freshly allocated outputs out, the
    launch on torch's current stream - and nothing for the CPU key: there is no CPU path;
  * a fake (meta) implementation, so the ops trace under FakeTensor / torch.compile / AOT autograd;
  * `register_autograd` where the reference differentiates through the function (raw2outputs RN:262-305 -> MLP parameters,
    gauss_net's gather GN:53-119 -> the perturbation).
`torch.library.opcheck` runs on every op in tests/test_hip_ops.py. The module-level functions of the mirrors (raw2outputs,
create_gauss_w, igsm_step, knn8, get_rays ...) call these ops. SURVEY's `render_rays_fused_fwd/bwd` is their composition:
`_train.RenderRaysTrain` (a torch.autograd.Function over ray sampling, mlp_fwd_train, composite, sample_fine and, backwards,
composite_bwd and mlp_bwd), selected by render_rays whenever a NeRF parameter requires grad."""
import torch
from torch import Tensor
from torch.library import custom_op

from . import _lib

NS = 'nerfail_mi'


def _chk(rc):
    _lib.check(rc)


def _s():
    return _lib.stream()


# ----------------------------------------------------------------------------------------------- K1 rays (RH:157-166, RN:102-123)
@custom_op(NS + '::ray_gen', mutates_args=(), device_types='cuda')
def ray_gen(K: Tensor, c2w: Tensor, anchor: Tensor, H: int, W: int, near: float, far: float, pix_begin: int, pix_count: int) -> Tensor:
    """Packed rays [pix_count, 11] (o, d, near, far, viewdir) of pixels [pix_begin, pix_begin + pix_count) of an H x W view.
    K [3,3] and c2w [3,4] may live on the host (12 + 4 floats are passed by value); `anchor` is any tensor on the target device."""
    k = K.detach().to('cpu', torch.float64)
    c = c2w.detach().to('cpu', torch.float32)[:3, :4].reshape(-1).tolist()
    rays = torch.empty((pix_count, _lib.RAY_FLOATS), dtype=torch.float32, device=anchor.device)
    _chk(_lib.load().nerfail_ray_gen(H, W, _lib.host_floats([k[0, 0], k[1, 1], k[0, 2], k[1, 2]]), _lib.host_floats(c), near, far,
                                     pix_begin, pix_count, _lib.dev(rays), _s()))
    return rays


@ray_gen.register_fake
def _(K, c2w, anchor, H, W, near, far, pix_begin, pix_count):
    return anchor.new_empty((pix_count, _lib.RAY_FLOATS), dtype=torch.float32)


# ----------------------------------------------------------------------------------------------- K5 + K7 composite (RN:262-305)
@custom_op(NS + '::composite', mutates_args=(), device_types='cuda')
def composite(raw: Tensor, z_vals: Tensor, rays: Tensor, white_bkgd: bool) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """raw2outputs: (rgb_map [R,3], disp_map [R], acc_map [R], weights [R,N], depth_map [R])."""
    R, N = z_vals.shape
    dev = raw.device
    o = [torch.empty(s, dtype=torch.float32, device=dev) for s in ((R, 3), (R,), (R,), (R, N), (R,))]
    _chk(_lib.load().nerfail_composite(_lib.dev(raw), _lib.dev(z_vals), _lib.dev(rays), None, R, N, int(white_bkgd), _lib.dev(o[0]),
                                       _lib.dev(o[1]), _lib.dev(o[2]), _lib.dev(o[3]), _lib.dev(o[4]), None, None, _s()))
    return o[0], o[1], o[2], o[3], o[4]


@composite.register_fake
def _(raw, z_vals, rays, white_bkgd):
    R, N = z_vals.shape
    return (raw.new_empty((R, 3)), raw.new_empty((R,)), raw.new_empty((R,)), raw.new_empty((R, N)), raw.new_empty((R,)))


@custom_op(NS + '::composite_bwd', mutates_args=(), device_types='cuda')
def composite_bwd(raw: Tensor, z_vals: Tensor, rays: Tensor, white_bkgd: bool, g_rgb: Tensor, g_disp: Tensor, g_acc: Tensor,
                  g_weights: Tensor, g_depth: Tensor) -> Tensor:
    """d(loss)/d(raw) for upstream gradients of all five outputs (autograd of RN:262-305)."""
    R, N = z_vals.shape
    d_raw = torch.empty((R, N, 4), dtype=torch.float32, device=raw.device)
    _chk(_lib.load().nerfail_composite_bwd(_lib.dev(raw), _lib.dev(z_vals), _lib.dev(rays), None, R, N, int(white_bkgd),
                                           _lib.dev(g_rgb), _lib.dev(g_disp), _lib.dev(g_acc), _lib.dev(g_depth), _lib.dev(g_weights),
                                           _lib.dev(d_raw), _s()))
    return d_raw


@composite_bwd.register_fake
def _(raw, z_vals, rays, white_bkgd, g_rgb, g_disp, g_acc, g_weights, g_depth):
    return raw.new_empty(raw.shape)


def _composite_setup(ctx, inputs, output):
    raw, z_vals, rays, white = inputs
    ctx.save_for_backward(raw, z_vals, rays)
    ctx.white = white


def _composite_backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
    raw, z_vals, rays = ctx.saved_tensors
    R, N = z_vals.shape

    def z(g, shape):
        return torch.zeros(shape, dtype=torch.float32, device=raw.device) if g is None else g.contiguous().float()
    d_raw = composite_bwd(raw, z_vals, rays, ctx.white, z(g_rgb, (R, 3)), z(g_disp, (R,)), z(g_acc, (R,)), z(g_w, (R, N)), z(g_depth, (R,)))
    return d_raw, None, None, None


composite.register_autograd(_composite_backward, setup_context=_composite_setup)


# ----------------------------------------------------------------------------------------------- K6 sample_pdf (RH:200-243)
@custom_op(NS + '::sample_pdf', mutates_args=(), device_types='cuda')
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF samples [R, n]; u is [n] (shared, det=True) or [R, n] (explicit draws)."""
    R, nb = bins.shape
    n = u.shape[-1]
    out = torch.empty((R, n), dtype=torch.float32, device=bins.device)
    _chk(_lib.load().nerfail_sample_pdf(_lib.dev(bins), _lib.dev(weights), R, nb, _lib.dev(u), int(u.dim() == 1), n, _lib.dev(out), _s()))
    return out


@sample_pdf.register_fake
def _(bins, weights, u):
    return bins.new_empty((bins.shape[0], u.shape[-1]))


# ----------------------------------------------------------------------------------------------- K3 + K4 fused encode + MLP (RN:37-51)
@custom_op(NS + '::mlp_fwd', mutates_args=(), device_types='cuda')
def mlp_fwd(packed: Tensor, pts: Tensor, viewdirs: Tensor, D: int, W: int, skip: int) -> Tensor:
    """raw [R,N,4] of the NeRF MLP on points [R,N,3] with per-ray view directions [R,3] (weights: NeRF.packed())."""
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=pts.device)
    _chk(_lib.load().nerfail_mlp_fwd(_lib.dev(packed), D, W, skip, _lib.dev(pts), _lib.dev(viewdirs), R * N, N, _lib.dev(raw), _s()))
    return raw


@mlp_fwd.register_fake
def _(packed, pts, viewdirs, D, W, skip):
    return pts.new_empty((pts.shape[0], pts.shape[1], 4))


# ----------------------------------------------------------------------------------------------- K8 exact 8-NN (CI:126-145)
@custom_op(NS + '::knn8', mutates_args=(), device_types='cuda')
def knn8(queries: Tensor, points: Tensor) -> tuple[Tensor, Tensor]:
    """(dist [Q,8] ascending, idx [Q,8] as float32 - the on-disk convention, CI:148-163) of queries [Q,3] in points [M,3]."""
    lib = _lib.load()
    Q, M = queries.shape[0], points.shape[0]
    dist = torch.empty((Q, 8), dtype=torch.float32, device=queries.device)
    idx = torch.empty((Q, 8), dtype=torch.float32, device=queries.device)
    if M >= 4096:
        nb = lib.nerfail_knn8_grid_workspace_bytes(M)
        ws = torch.empty((nb,), dtype=torch.uint8, device=queries.device)
        _chk(lib.nerfail_knn8_grid(_lib.dev(queries), Q, _lib.dev(points), M, _lib.dev(dist), _lib.dev(idx), None, _lib.dev(ws), nb, _s()))
    else:
        _chk(lib.nerfail_knn8(_lib.dev(queries), Q, _lib.dev(points), M, _lib.dev(dist), _lib.dev(idx), None, _s()))
    return dist, idx


@knn8.register_fake
def _(queries, points):
    return queries.new_empty((queries.shape[0], 8)), queries.new_empty((queries.shape[0], 8))


# ----------------------------------------------------------------------------------------------- K9 create_gauss_w (GN:169-186)
@custom_op(NS + '::gauss_weight', mutates_args=(), device_types='cuda')
def gauss_weight(dist_and_index: Tensor, c: float) -> Tensor:
    B, P = dist_and_index.shape[0], dist_and_index.shape[2] * dist_and_index.shape[3]
    out = torch.empty_like(dist_and_index)
    _chk(_lib.load().nerfail_gauss_weight(_lib.dev(dist_and_index), B, P, c, _lib.dev(out), _s()))
    return out


@gauss_weight.register_fake
def _(dist_and_index, c):
    return torch.empty_like(dist_and_index)


# ----------------------------------------------------------------------------------------------- K10 / K11 gather (GN:53-119)
@custom_op(NS + '::gauss_gather', mutates_args=(), device_types='cuda')
def gauss_gather(spatial: Tensor, weight_and_index: Tensor, ori_img: Tensor, epsilon: float) -> tuple[Tensor, Tensor]:
    """(x, x_rgba) [B,H,W,4]; epsilon < 0 = no clip (epsilon=None in the reference)."""
    wi = weight_and_index
    B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
    s = spatial.reshape(-1, 4)
    x = torch.empty(ori_img.shape, dtype=torch.float32, device=s.device)
    xr = torch.empty(ori_img.shape, dtype=torch.float32, device=s.device)
    _chk(_lib.load().nerfail_gauss_fwd(_lib.dev(s), s.shape[0], _lib.dev(wi), _lib.dev(ori_img), B, P, epsilon, _lib.dev(x),
                                       _lib.dev(xr), None, _s()))
    return x, xr


@gauss_gather.register_fake
def _(spatial, weight_and_index, ori_img, epsilon):
    return torch.empty_like(ori_img), torch.empty_like(ori_img)


@custom_op(NS + '::gauss_gather_bwd', mutates_args=(), device_types='cuda')
def gauss_gather_bwd(weight_and_index: Tensor, ori_img: Tensor, x: Tensor, grad_x: Tensor, grad_x_rgba: Tensor, n_rows: int,
                     epsilon: float) -> Tensor:
    """d/d(spatial) [n_rows,4]: the stateless (float-atomic) form of K11. The attack loop's deterministic form with cached
    per-view indices is nerfail_amd.GaussNet.gauss_gather / nerfail_gauss_bwd_views."""
    wi = weight_and_index
    B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
    gs = torch.zeros((n_rows, 4), dtype=torch.float32, device=x.device)
    _chk(_lib.load().nerfail_gauss_bwd(_lib.dev(wi), _lib.dev(ori_img), _lib.dev(x), _lib.dev(grad_x), _lib.dev(grad_x_rgba), n_rows,
                                       B, P, epsilon, _lib.dev(gs), _s()))
    return gs


@gauss_gather_bwd.register_fake
def _(weight_and_index, ori_img, x, grad_x, grad_x_rgba, n_rows, epsilon):
    return x.new_empty((n_rows, 4))


def _gather_setup(ctx, inputs, output):
    spatial, wi, ori, eps = inputs
    ctx.save_for_backward(wi, ori, output[0])
    ctx.eps, ctx.s_shape = eps, spatial.shape


def _gather_backward(ctx, g_x, g_xr):
    wi, ori, x = ctx.saved_tensors

    def z(g):
        return torch.zeros_like(x) if g is None else g.contiguous().float()
    n = 1
    for d in ctx.s_shape[:-1]:
        n *= d
    return gauss_gather_bwd(wi, ori, x, z(g_x), z(g_xr), n, ctx.eps).reshape(ctx.s_shape), None, None, None


gauss_gather.register_autograd(_gather_backward, setup_context=_gather_setup)


# ----------------------------------------------------------------------------------------------- K12 sign step (AS:352-392)
@custom_op(NS + '::igsm_step', mutates_args=(), device_types='cuda')
def igsm_step(spatial: Tensor, grad: Tensor, spatial_init: Tensor, a: float, epsilon: float, targeted: bool) -> Tensor:
    out = torch.empty_like(spatial)
    _chk(_lib.load().nerfail_igsm_step(_lib.dev(spatial), _lib.dev(grad), _lib.dev(spatial_init), spatial.numel() // 4, a, epsilon,
                                       int(targeted), _lib.dev(out), _s()))
    return out


@igsm_step.register_fake
def _(spatial, grad, spatial_init, a, epsilon, targeted):
    return torch.empty_like(spatial)


# ----------------------------------------------------------------------------------------------- K6 fused fine sampling (RN:392-412)
@custom_op(NS + '::sample_fine', mutates_args=(), device_types='cuda')
def sample_fine(rays: Tensor, z_coarse: Tensor, weights: Tensor, u: Tensor) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """sample_pdf on the mid-points (RH:200-243) + sort(cat(z_vals, z_samples)) (RN:397: SURVEY's `merge_sorted`) +
    pts = o + d z (RN:399) + z_std (RN:412) in one launch: (z_samples [R,nf], z_fine [R,nc+nf], pts [R,nc+nf,3], z_std [R]).
    u is [nf] (shared, perturb = 0) or [R,nf] (explicit draws)."""
    R, nc = z_coarse.shape
    nf = u.shape[-1]
    dev = z_coarse.device
    zs = torch.empty((R, nf), dtype=torch.float32, device=dev)
    zf = torch.empty((R, nc + nf), dtype=torch.float32, device=dev)
    pts = torch.empty((R, nc + nf, 3), dtype=torch.float32, device=dev)
    zstd = torch.empty((R,), dtype=torch.float32, device=dev)
    _chk(_lib.load().nerfail_sample_fine(_lib.dev(rays), R, _lib.dev(z_coarse), _lib.dev(weights), nc, _lib.dev(u), int(u.dim() == 1), nf,
                                         _lib.dev(zs), _lib.dev(zf), _lib.dev(pts), _lib.dev(zstd), _s()))
    return zs, zf, pts, zstd


@sample_fine.register_fake
def _(rays, z_coarse, weights, u):
    R, nc = z_coarse.shape
    nf = u.shape[-1]
    return z_coarse.new_empty((R, nf)), z_coarse.new_empty((R, nc + nf)), z_coarse.new_empty((R, nc + nf, 3)), z_coarse.new_empty((R,))


# ----------------------------------------------------------------------------------------------- K4b training kernels (RN:776-801)
@custom_op(NS + '::mlp_fwd_train', mutates_args=(), device_types='cuda')
def mlp_fwd_train(packed: Tensor, pts: Tensor, viewdirs: Tensor, D: int, W: int, skip: int) -> tuple[Tensor, Tensor]:
    """mlp_fwd that also saves what the backward needs: (raw [R,N,4], acts [nerfail_mlp_train_acts_floats])."""
    lib = _lib.load()
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=pts.device)
    acts = torch.empty((lib.nerfail_mlp_train_acts_floats(D, W, R * N),), dtype=torch.float32, device=pts.device)
    _chk(lib.nerfail_mlp_fwd_train(_lib.dev(packed), D, W, skip, _lib.dev(pts), _lib.dev(viewdirs), R * N, N, _lib.dev(raw), _lib.dev(acts), _s()))
    return raw, acts


@mlp_fwd_train.register_fake
def _(packed, pts, viewdirs, D, W, skip):
    M = pts.shape[0] * pts.shape[1]
    return pts.new_empty((pts.shape[0], pts.shape[1], 4)), pts.new_empty((_lib.load().nerfail_mlp_train_acts_floats(D, W, M),))


def mlp_param_shapes(D, W, skip, input_ch=63, input_ch_views=27):
    """Shapes of NeRF's parameters in _train.ordered_params order (RH:72-98): pts_linears (w, b) x D, views, feature, alpha, rgb."""
    shapes = []
    for i in range(D):
        fan_in = input_ch if i == 0 else (W + input_ch if (skip >= 0 and i == skip + 1) else W)
        shapes += [(W, fan_in), (W,)]
    return shapes + [(W // 2, W + input_ch_views), (W // 2,), (W, W), (W,), (1, W), (1,), (3, W // 2), (3,)]


@custom_op(NS + '::mlp_bwd', mutates_args=(), device_types='cuda')
def mlp_bwd(packed: Tensor, packed_T: Tensor, acts: Tensor, d_raw: Tensor, D: int, W: int, skip: int) -> list[Tensor]:
    """d loss / d (NeRF parameters) from d loss / d raw [R,N,4] and the activations mlp_fwd_train saved: backward-data pass
    (layer gradients) + weight-gradient pass. Returns the 2 D + 8 gradients in _train.ordered_params order.
    packed_T = _train.packed_T(net) (nerfail_mlp_pack_T)."""
    lib = _lib.load()
    M = d_raw.shape[0] * d_raw.shape[1]
    dev = d_raw.device
    dz = torch.empty((lib.nerfail_mlp_train_dz_floats(D, W, M),), dtype=torch.float32, device=dev)
    _chk(lib.nerfail_mlp_bwd_data(_lib.dev(packed), _lib.dev(packed_T), D, W, skip, _lib.dev(d_raw), _lib.dev(acts), M, _lib.dev(dz), _s()))
    grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in mlp_param_shapes(D, W, skip)]     # overwritten
    mp = _lib.MlpParams()
    mp.D, mp.W, mp.input_ch, mp.input_ch_views, mp.skip = D, W, 63, 27, skip
    it = iter(grads)
    for i in range(D):
        mp.pts_w[i], mp.pts_b[i] = next(it).data_ptr(), next(it).data_ptr()
    mp.views_w, mp.views_b = next(it).data_ptr(), next(it).data_ptr()
    mp.feature_w, mp.feature_b = next(it).data_ptr(), next(it).data_ptr()
    mp.alpha_w, mp.alpha_b = next(it).data_ptr(), next(it).data_ptr()
    mp.rgb_w, mp.rgb_b = next(it).data_ptr(), next(it).data_ptr()
    nbytes = lib.nerfail_mlp_bwd_weights_scratch_bytes(D, W, skip, M, 0, 0)
    scratch = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)      # per-workgroup partials (deterministic sum)
    _chk(lib.nerfail_mlp_bwd_weights(D, W, skip, _lib.dev(acts), _lib.dev(dz), M, mp, 0, None, 0, _lib.dev(scratch), nbytes, _s()))
    return grads


@mlp_bwd.register_fake
def _(packed, packed_T, acts, d_raw, D, W, skip):
    return [d_raw.new_empty(sh) for sh in mlp_param_shapes(D, W, skip)]


ALL = ('ray_gen', 'composite', 'composite_bwd', 'sample_pdf', 'sample_fine', 'mlp_fwd', 'mlp_fwd_train', 'mlp_bwd', 'knn8', 'gauss_weight',
       'gauss_gather', 'gauss_gather_bwd', 'igsm_step')
