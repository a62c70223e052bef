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


def packed_both(net):
    """Forward and transposed weight images of `net` from ONE launch (nerfail_mlp_pack_train), filling the caches of
    NeRF.packed() and packed_T(): the training loop re-packs both after every optimizer step."""
    params = list(net.parameters())
    key = tuple((p.data_ptr(), p._version) for p in params)
    keyT = tuple((p.data_ptr(), p._version) for p in ordered_params(net))
    if net._packed is not None and net._packed_key == key and getattr(net, '_packedT', None) is not None and net._packedT_key == keyT:
        return net._packed, net._packedT
    lib = _lib.load()
    n, nT = lib.nerfail_mlp_packed_floats(net.D, net.W, net._skip()), lib.nerfail_mlp_packed_T_floats(net.D, net.W, net._skip())
    if n == 0:
        raise NotImplementedError('unsupported NeRF shape D=%d W=%d (W in {64,128,256})' % (net.D, net.W))
    keep = []
    mp = net._mlp_params(keep)
    buf = torch.empty((n,), dtype=torch.float32, device=params[0].device)
    bufT = torch.empty((nT,), dtype=torch.float32, device=params[0].device)
    _lib.check(lib.nerfail_mlp_pack_train(mp, _lib.dev(buf), _lib.dev(bufT), _lib.stream()))
    net._packed, net._packed_key = buf, key
    net._packedT, net._packedT_key = bufT, keyT
    return buf, bufT


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


def acts_floats(net, M):
    return _lib.load().nerfail_mlp_train_acts_floats(net.D, net.W, M)


def dz_floats(net, M):
    return _lib.load().nerfail_mlp_train_dz_floats(net.D, net.W, M)


def mlp_fwd_train(net, pts, viewdirs, acts=None):
    """Forward that saves the activations; `acts`: where (a slice of a buffer shared by the coarse and the fine pass, so
    that ONE backward launch can walk both), else a fresh buffer."""
    lib = _lib.load()
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=pts.device)
    if acts is None:
        acts = torch.empty((acts_floats(net, R * N),), dtype=torch.float32, device=pts.device)
    elif acts.numel() != acts_floats(net, R * N):
        raise ValueError('acts buffer has %d floats, the pass needs %d' % (acts.numel(), acts_floats(net, R * N)))
    if getattr(net, 'precision', 'f32') != 'f16x3':
        packed_both(net)                                     # one launch for both images (the backward needs the second)
    if getattr(net, 'precision', 'f32') == 'f16x3':        # split-precision forward, same saved activations
        _lib.check(lib.nerfail_mlp_fwd_f16_train(_lib.dev(net.packed()), _lib.dev(net.packed_f16()), net.D, net.W, net._skip(),
                                                 _lib.dev(pts), _lib.dev(viewdirs), R * N, N, _lib.dev(raw), _lib.dev(acts),
                                                 _lib.stream()))
        return raw, acts
    _lib.check(lib.nerfail_mlp_fwd_train(_lib.dev(net.packed()), net.D, net.W, net._skip(), _lib.dev(pts), _lib.dev(viewdirs),
                                         R * N, N, _lib.dev(raw), _lib.dev(acts), _lib.stream()))
    return raw, acts


def mlp_fwd_train_rays(net, rays, z_vals, acts=None):
    """mlp_fwd_train with the sample points formed inside the kernel (nerfail_mlp_fwd_rays): rays [R,11], z_vals [R,N]."""
    R, N = z_vals.shape
    if acts is None:
        acts = torch.empty((acts_floats(net, R * N),), dtype=torch.float32, device=z_vals.device)
    elif acts.numel() != acts_floats(net, R * N):
        raise ValueError('acts buffer has %d floats, the pass needs %d' % (acts.numel(), acts_floats(net, R * N)))
    packed_both(net)
    raw = torch.empty((R, N, 4), dtype=torch.float32, device=z_vals.device)
    _lib.check(_lib.load().nerfail_mlp_fwd_rays(_lib.dev(net.packed()), net.D, net.W, net._skip(), _lib.dev(rays), _lib.dev(z_vals), R, N,
                                                _lib.dev(raw), _lib.dev(acts), _lib.stream()))
    return raw, acts


def _same_arch(a, b):
    return (a.D, a.W, a._skip(), getattr(a, 'precision', 'f32')) == (b.D, b.W, b._skip(), getattr(b, 'precision', 'f32'))


def dw_scratch(net, M0, M1, flags, device):
    n = _lib.load().nerfail_mlp_bwd_weights_scratch_bytes(net.D, net.W, net._skip(), M0, M1, flags)
    return torch.empty((max(n, 1),), dtype=torch.uint8, device=device), n


def mlp_backward(net, d_raw, acts, grads, accumulate=True):
    """d loss / d params of ONE network into `grads` (list of tensors in ordered_params order; += when `accumulate`)."""
    mlp_backward2(net, d_raw.reshape(-1, 4), acts, grads, d_raw.shape[0] * d_raw.shape[1], None, None, 0, accumulate)


def mlp_backward2(net0, d_raw, acts, grads0, M0, net1, grads1, M1, accumulate=False):
    """Backward-data + weight gradients of one or TWO networks of the same architecture in ONE launch each: d_raw [M0+M1,4],
    acts hold network 0's tiles then network 1's. The W = 256 weight-gradient kernel is deterministic (no atomics)."""
    lib = _lib.load()
    M = M0 + M1
    split = getattr(net0, 'precision', 'f32') == 'f16x3'
    dz = torch.empty((dz_floats(net0, M0) + (dz_floats(net1, M1) if M1 else 0),), dtype=torch.float32, device=d_raw.device)
    st = _lib.stream()
    if split:
        off_a = off_z = off_r = 0
        for net, Mi in ((net0, M0), (net1, M1)):
            if Mi == 0:
                continue
            _lib.check(lib.nerfail_mlp_bwd_data_f16(_lib.dev(net.packed()), _lib.dev(packed_f16_T(net)), net.D, net.W, net._skip(),
                                                    _lib.dev(d_raw[off_r:off_r + Mi]), _lib.dev(acts[off_a:off_a + acts_floats(net, Mi)]), Mi,
                                                    _lib.dev(dz[off_z:off_z + dz_floats(net, Mi)]), st))
            off_a, off_z, off_r = off_a + acts_floats(net, Mi), off_z + dz_floats(net, Mi), off_r + Mi
    else:
        p0, pT0 = packed_both(net0)
        p1, pT1 = packed_both(net1) if M1 else (None, None)
        _lib.check(lib.nerfail_mlp_bwd_data2(_lib.dev(p0), _lib.dev(pT0), M0, _lib.dev(p1), _lib.dev(pT1), M1, net0.D, net0.W,
                                             net0._skip(), _lib.dev(d_raw), _lib.dev(acts), _lib.dev(dz), st))
    flags = (_lib.DW_BF16X3 if split else 0) | (_lib.DW_ACCUMULATE if accumulate else 0)
    scratch, nbytes = dw_scratch(net0, M0, M1, flags, d_raw.device)
    _lib.check(lib.nerfail_mlp_bwd_weights(net0.D, net0.W, net0._skip(), _lib.dev(acts), _lib.dev(dz), M0, _grads_struct(net0, grads0),
                                           M1, _grads_struct(net1, grads1) if M1 else None, flags, _lib.dev(scratch), nbytes, st))


def composite_backward(raw, z_vals, rays, noise, white_bkgd, g_rgb, g_disp, g_acc, g_depth=None, g_weights=None, out=None):
    R, N = z_vals.shape
    d_raw = torch.empty((R, N, 4), dtype=torch.float32, device=raw.device) if out is None else out.view(R, N, 4)

    def c(t):
        return None if t is None else _lib.f32c(t)
    g_rgb, g_disp, g_acc, g_depth, g_weights = c(g_rgb), c(g_disp), c(g_acc), c(g_depth), c(g_weights)
    _lib.check(_lib.load().nerfail_composite_bwd(_lib.dev(raw), _lib.dev(z_vals), _lib.dev(rays), _lib.dev(noise), R, N,
                                                 int(bool(white_bkgd)), _lib.dev(g_rgb), _lib.dev(g_disp), _lib.dev(g_acc),
                                                 _lib.dev(g_depth), _lib.dev(g_weights), _lib.dev(d_raw), _lib.stream()))
    return d_raw


def _new_grads(net, zero):
    """One flat buffer per network, sliced into per-parameter views (`zero`: one memset; the W = 256 weight-gradient
    kernel OVERWRITES its outputs, so the training step needs none)."""
    ps = ordered_params(net)
    n = sum(p.numel() for p in ps)
    flat = (torch.zeros if zero else torch.empty)((n,), dtype=torch.float32, device=ps[0].device)
    out, off = [], 0
    for p in ps:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


def _zero_grads(net):
    return _new_grads(net, True)


class RenderRaysTrain(torch.autograd.Function):
    """forward(rays, cfg, *params) -> (rgb_map, disp_map, acc_map, rgb0, disp0, acc0, z_std, pts_max, raw)."""

    @staticmethod
    def forward(ctx, rays, cfg, *params):
        out = cfg['pipeline'](rays, train=True)
        ctx.cfg = cfg
        ctx.saved = out['_saved']
        ctx.mark_non_differentiable(out['z_std'], out['pts_max'])
        ctx.set_materialize_grads(False)          # unused outputs arrive as None, not as freshly filled zero tensors
        return (out['rgb_map'], out['disp_map'], out['acc_map'], out['rgb0'], out['disp0'], out['acc0'], out['z_std'],
                out['pts_max'], out['raw'])

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_rgb0, g_disp0, g_acc0, g_zstd, g_ptsmax, g_raw):
        cfg, sv = ctx.cfg, ctx.saved
        coarse, fine = cfg['network_fn'], cfg['network_fine']
        nets = [coarse] + ([fine] if fine is not None else [])
        wb = cfg['white_bkgd']
        c, f = sv['coarse'], sv['fine']
        Mc = c['z'].shape[0] * c['z'].shape[1]
        if f is None:
            d_raw = composite_backward(c['raw'], c['z'], sv['rays'], c['noise'], wb, g_rgb, g_disp, g_acc)
            if g_raw is not None and cfg['retraw']:
                d_raw = d_raw + g_raw
            grads = {id(coarse): _new_grads(coarse, False)}
            mlp_backward2(coarse, d_raw.reshape(-1, 4), c['acts'], grads[id(coarse)], Mc, None, None, 0)
        else:
            run = fine if fine is not None else coarse
            Mf = f['z'].shape[0] * f['z'].shape[1]
            joint = sv.get('acts_all') is not None and Mc % 32 == 0 and _same_arch(coarse, run)
            d_all = torch.empty((Mc + Mf, 4), dtype=torch.float32, device=c['raw'].device)
            composite_backward(c['raw'], c['z'], sv['rays'], c['noise'], wb, g_rgb0, g_disp0, g_acc0, out=d_all[:Mc])
            composite_backward(f['raw'], f['z'], sv['rays'], f['noise'], wb, g_rgb, g_disp, g_acc, out=d_all[Mc:])
            if g_raw is not None and cfg['retraw']:
                d_all[Mc:] += g_raw.reshape(-1, 4)
            grads = {id(n): _new_grads(n, not joint) for n in nets}
            if joint and run is coarse:            # one network evaluated twice (network_fine=None): one run of Mc + Mf samples
                mlp_backward2(coarse, d_all, sv['acts_all'], grads[id(coarse)], Mc + Mf, None, None, 0)
            elif joint:                            # coarse + fine in ONE launch each (independent: RN:394 detaches z_samples)
                mlp_backward2(coarse, d_all, sv['acts_all'], grads[id(coarse)], Mc, run, grads[id(run)], Mf)
            else:
                mlp_backward2(run, d_all[Mc:], f['acts'], grads[id(run)], Mf, None, None, 0, accumulate=True)
                mlp_backward2(coarse, d_all[:Mc], c['acts'], grads[id(coarse)], Mc, None, None, 0, accumulate=True)
        flat = []
        for n in nets:
            flat += grads[id(n)]
        return (None, None) + tuple(flat)
