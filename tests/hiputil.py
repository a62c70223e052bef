"""Helpers shared by the -m gpu parity tests (HIP path through the C ABI vs oracle / golden)."""
import numpy as np
import torch

import synth


def dev():
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


def hip_nerf(D=8, W=256, seed=0, requires_grad=False, precision='f32'):
    from nerfail_amd.run_nerf_helpers import NeRF
    sd = synth.nerf_state_dict(D=D, W=W, seed=seed)
    m = NeRF(D=D, W=W, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.requires_grad_(requires_grad)       # forward tests: inference path; training tests ask for gradients
    m.precision = precision
    return sd, m.to(dev())


def torch_nerf_mlp(sd, pts, dirs, dtype):
    """Plain torch NeRF MLP (D=8, W=256, skip 4, use_viewdirs) on flat points/dirs in `dtype`; returns raw [M,4] and the
    leaf parameters (float64 = the ground truth the fp32 / split-precision gradient kernels are judged against)."""
    P = {k: torch.from_numpy(v).to(pts.device, dtype).requires_grad_(True) for k, v in sd.items()}

    def emb(x, L):
        out = [x]
        for f in range(L):
            out += [torch.sin(x * 2. ** f), torch.cos(x * 2. ** f)]
        return torch.cat(out, -1)
    e, ed = emb(pts.to(dtype), 10), emb(dirs.to(dtype), 4)
    h = e
    for i in range(8):
        h = torch.relu(h @ P['pts_linears.%d.weight' % i].T + P['pts_linears.%d.bias' % i])
        if i == 4:
            h = torch.cat([e, h], -1)
    alpha = h @ P['alpha_linear.weight'].T + P['alpha_linear.bias']
    feat = h @ P['feature_linear.weight'].T + P['feature_linear.bias']
    h = torch.relu(torch.cat([feat, ed], -1) @ P['views_linears.0.weight'].T + P['views_linears.0.bias'])
    rgb = h @ P['rgb_linear.weight'].T + P['rgb_linear.bias']
    return torch.cat([rgb, alpha], -1), P


def hip_mlp_grads(net, pts, dirs, d_raw, fwd='f32', bd='f32', dw='f32'):
    """Parameter gradients of sum(raw * d_raw) through the C ABI with each of the three kernel families chosen
    separately ('f32' | 'split'): forward-with-activations, backward-data, weight gradients. pts [R,N,3], dirs [R,3]."""
    from nerfail_amd import _lib, _train
    lib = _lib.load()
    R, N_ = pts.shape[0], pts.shape[1]
    M = R * N_
    net.precision = 'f16x3' if fwd == 'split' else 'f32'
    raw, acts = _train.mlp_fwd_train(net, pts, dirs)
    g = _train._zero_grads(net)
    dz = torch.empty((lib.nerfail_mlp_train_dz_floats(net.D, net.W, M),), dtype=torch.float32, device=d_raw.device)
    if bd == 'split':
        _lib.check(lib.nerfail_mlp_bwd_data_f16(_lib.dev(net.packed()), _lib.dev(_train.packed_f16_T(net)), net.D, net.W,
                                                net._skip(), _lib.dev(d_raw), _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))
    else:
        _lib.check(lib.nerfail_mlp_bwd_data(_lib.dev(net.packed()), _lib.dev(_train.packed_T(net)), net.D, net.W,
                                            net._skip(), _lib.dev(d_raw), _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))
    flags = (_lib.DW_BF16X3 if dw == 'split' else 0) | _lib.DW_ACCUMULATE
    scratch, nbytes = _train.dw_scratch(net, M, 0, flags, d_raw.device)
    _lib.check(lib.nerfail_mlp_bwd_weights(net.D, net.W, net._skip(), _lib.dev(acts), _lib.dev(dz), M, _train._grads_struct(net, g), 0, None,
                                           flags, _lib.dev(scratch), nbytes, _lib.stream()))
    byp = {id(p): n for n, p in net.named_parameters()}
    return {byp[id(p)]: t.double().cpu().numpy() for p, t in zip(_train.ordered_params(net), g)}
