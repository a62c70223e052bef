"""Autograd wiring of the training step (run_nerf.py:776-791): render_rays as ONE torch.autograd.Function whose
forward runs the fused HIP pipeline (saving activations in register-fragment layout) and whose backward runs
nerfail_composite_bwd -> nerfail_mlp_bwd_data -> nerfail_mlp_bwd_weights for the fine and the coarse pass.

Gradients flow to the parameters of network_fn / network_fine only: rays are data and z_samples is detached
(RN:394), exactly the graph the reference's loss.backward() sees."""
import torch

from . import _lib

PARAM_ORDER = None


def ordered_params(net):
    """Parameters in the fixed order used for Function.apply / grads: pts_linears (w, b)*D, views, feature, alpha, rgb."""
    ps = []
    for l in net.pts_linears:
        ps += [l.weight, l.bias]
    ps += [net.views_linears[0].weight, net.views_linears[0].bias, net.feature_linear.weight, net.feature_linear.bias,
           net.alpha_linear.weight, net.alpha_linear.bias, net.rgb_linear.weight, net.rgb_linear.bias]
    return ps


def _grads_struct(net, tensors):
    """nerfail_mlp_params filled with the data pointers of `tensors` (same order as ordered_params)."""
    mp = _lib.MlpParams()
    mp.D, mp.W, mp.input_ch, mp.input_ch_views, mp.skip = net.D, net.W, net.input_ch, net.input_ch_views, net._skip()
    it = iter(tensors)
    for i in range(net.D):
        mp.pts_w[i] = next(it).data_ptr()
        mp.pts_b[i] = next(it).data_ptr()
    mp.views_w, mp.views_b = next(it).data_ptr(), next(it).data_ptr()
    mp.feature_w, mp.feature_b = next(it).data_ptr(), next(it).data_ptr()
    mp.alpha_w, mp.alpha_b = next(it).data_ptr(), next(it).data_ptr()
    mp.rgb_w, mp.rgb_b = next(it).data_ptr(), next(it).data_ptr()
    return mp


def packed_T(net):
    """Transposed weight image for the backward-data pass, cached like NeRF.packed()."""
    params = ordered_params(net)
    key = tuple((p.data_ptr(), p._version) for p in params)
    if getattr(net, '_packedT', None) is not None and net._packedT_key == key:
        return net._packedT
    lib = _lib.load()
    n = lib.nerfail_mlp_packed_T_floats(net.D, net.W, net._skip())
    keep = [_lib.f32c(p) for p in params]
    mp = _grads_struct(net, keep)
    buf = torch.empty((n,), dtype=torch.float32, device=params[0].device)
    _lib.check(lib.nerfail_mlp_pack_T(mp, _lib.dev(buf), _lib.stream()))
    net._packedT, net._packedT_key = buf, key
    return buf


def packed_f16_T(net):
    """fp16 hi/lo image of the transposed weights (split-precision backward-data), cached on parameter versions."""
    params = ordered_params(net)
    key = tuple((p.data_ptr(), p._version) for p in params)
    if getattr(net, '_packed16T', None) is not None and net._packed16T_key == key:
        return net._packed16T
    lib = _lib.load()
    n = lib.nerfail_mlp_f16_image_T_bytes(net.D, net.W, net._skip())
    net.check_f16x3_range()
    keep = [_lib.f32c(p) for p in params]
    mp = _grads_struct(net, keep)
    buf = torch.empty((n,), dtype=torch.uint8, device=params[0].device)
    _lib.check(lib.nerfail_mlp_pack_f16_T(mp, _lib.dev(buf), _lib.stream()))
    net._packed16T, net._packed16T_key = buf, key
    return buf


def mlp_fwd_train(net, pts, viewdirs):
    lib = _lib.load()
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=pts.device)
    acts = torch.empty((lib.nerfail_mlp_train_acts_floats(net.D, net.W, R * N),), dtype=torch.float32, device=pts.device)
    if getattr(net, 'precision', 'f32') == 'f16x3':        # split-precision forward, same saved activations
        _lib.check(lib.nerfail_mlp_fwd_f16_train(_lib.dev(net.packed()), _lib.dev(net.packed_f16()), net.D, net.W, net._skip(),
                                                 _lib.dev(pts), _lib.dev(viewdirs), R * N, N, _lib.dev(raw), _lib.dev(acts),
                                                 _lib.stream()))
        return raw, acts
    _lib.check(lib.nerfail_mlp_fwd_train(_lib.dev(net.packed()), net.D, net.W, net._skip(), _lib.dev(pts), _lib.dev(viewdirs),
                                         R * N, N, _lib.dev(raw), _lib.dev(acts), _lib.stream()))
    return raw, acts


def mlp_backward(net, d_raw, acts, grads):
    """Accumulates d loss / d params of `net` into `grads` (list of tensors in ordered_params order)."""
    lib = _lib.load()
    M = d_raw.shape[0] * d_raw.shape[1]
    dz = torch.empty((lib.nerfail_mlp_train_dz_floats(net.D, net.W, M),), dtype=torch.float32, device=d_raw.device)
    if getattr(net, 'precision', 'f32') == 'f16x3':
        _lib.check(lib.nerfail_mlp_bwd_data_f16(_lib.dev(net.packed()), _lib.dev(packed_f16_T(net)), net.D, net.W, net._skip(),
                                                _lib.dev(d_raw), _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))
    else:
        _lib.check(lib.nerfail_mlp_bwd_data(_lib.dev(net.packed()), _lib.dev(packed_T(net)), net.D, net.W, net._skip(),
                                            _lib.dev(d_raw), _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))
    fn = lib.nerfail_mlp_bwd_weights_bf16x3 if getattr(net, 'precision', 'f32') == 'f16x3' else lib.nerfail_mlp_bwd_weights
    _lib.check(fn(net.D, net.W, net._skip(), _lib.dev(acts), _lib.dev(dz), M, _grads_struct(net, grads), _lib.stream()))


def composite_backward(raw, z_vals, rays, noise, white_bkgd, g_rgb, g_disp, g_acc, g_depth=None, g_weights=None):
    R, N = z_vals.shape
    d_raw = torch.empty((R, N, 4), dtype=torch.float32, device=raw.device)

    def c(t):
        return None if t is None else _lib.f32c(t)
    g_rgb, g_disp, g_acc, g_depth, g_weights = c(g_rgb), c(g_disp), c(g_acc), c(g_depth), c(g_weights)
    _lib.check(_lib.load().nerfail_composite_bwd(_lib.dev(raw), _lib.dev(z_vals), _lib.dev(rays), _lib.dev(noise), R, N,
                                                 int(bool(white_bkgd)), _lib.dev(g_rgb), _lib.dev(g_disp), _lib.dev(g_acc),
                                                 _lib.dev(g_depth), _lib.dev(g_weights), _lib.dev(d_raw), _lib.stream()))
    return d_raw


def _zero_grads(net):
    ps = ordered_params(net)
    flat = torch.zeros((sum(p.numel() for p in ps),), dtype=torch.float32, device=ps[0].device)
    out, off = [], 0
    for p in ps:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


class RenderRaysTrain(torch.autograd.Function):
    """forward(rays, cfg, *params) -> (rgb_map, disp_map, acc_map, rgb0, disp0, acc0, z_std, pts_max, raw)."""

    @staticmethod
    def forward(ctx, rays, cfg, *params):
        out = cfg['pipeline'](rays, train=True)
        ctx.cfg = cfg
        ctx.saved = out['_saved']
        ctx.mark_non_differentiable(out['z_std'], out['pts_max'])
        return (out['rgb_map'], out['disp_map'], out['acc_map'], out['rgb0'], out['disp0'], out['acc0'], out['z_std'],
                out['pts_max'], out['raw'])

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_rgb0, g_disp0, g_acc0, g_zstd, g_ptsmax, g_raw):
        cfg, sv = ctx.cfg, ctx.saved
        coarse, fine = cfg['network_fn'], cfg['network_fine']
        nets = [coarse] + ([fine] if fine is not None else [])
        grads = {id(n): _zero_grads(n) for n in nets}      # one memset per network, sliced into per-parameter views
        wb = cfg['white_bkgd']
        if sv['fine'] is not None:
            f = sv['fine']
            d_raw = composite_backward(f['raw'], f['z'], sv['rays'], f['noise'], wb, g_rgb, g_disp, g_acc)
            if g_raw is not None and cfg['retraw']:
                d_raw = d_raw + g_raw
            run = fine if fine is not None else coarse
            mlp_backward(run, d_raw, f['acts'], grads[id(run)])
            c = sv['coarse']
            d_raw0 = composite_backward(c['raw'], c['z'], sv['rays'], c['noise'], wb, g_rgb0, g_disp0, g_acc0)
            mlp_backward(coarse, d_raw0, c['acts'], grads[id(coarse)])
        else:
            c = sv['coarse']
            d_raw0 = composite_backward(c['raw'], c['z'], sv['rays'], c['noise'], wb, g_rgb, g_disp, g_acc)
            if g_raw is not None and cfg['retraw']:
                d_raw0 = d_raw0 + g_raw
            mlp_backward(coarse, d_raw0, c['acts'], grads[id(coarse)])
        flat = []
        for n in nets:
            flat += grads[id(n)]
        return (None, None) + tuple(flat)
