"""-m gpu: the training step (row a12, run_nerf.py:776-791): loss.backward() through the HIP render path vs the
reference's autograd (fixture g7) and the float64-accumulating oracle."""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_err, l2_err
from hiputil import T, N, hip_nerf, dev
from oracle import nerf as O

pytestmark = pytest.mark.gpu


def _torch_raw2outputs(raw, z, rd, white):
    """float64 torch restatement of RN:262-305 used only to differentiate every output w.r.t. raw."""
    dists = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1) * rd.norm(dim=-1, keepdim=True)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1. - torch.exp(-torch.relu(raw[..., 3]) * dists)
    T_ = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T_
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth = (w * z).sum(-1)
    acc = w.sum(-1)
    disp = 1. / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    if white:
        rgb_map = rgb_map + (1. - acc[..., None])
    return rgb_map, disp, acc, w, depth


@pytest.mark.parametrize('Ns', [64, 192])
@pytest.mark.parametrize('white', [False, True])
def test_composite_backward_all_outputs(Ns, white):
    from nerfail_amd._train import composite_backward
    rs = np.random.RandomState(Ns)
    R = 37
    z = np.sort(rs.uniform(2, 6, (R, Ns)).astype(np.float32), -1)
    raw = rs.normal(size=(R, Ns, 4)).astype(np.float32)
    raw[..., 3] = raw[..., 3] * 2 + 0.5
    rd = rs.normal(size=(R, 3)).astype(np.float32)
    g = [rs.normal(size=s).astype(np.float32) for s in ((R, 3), (R,), (R,), (R, Ns), (R,))]
    rt = torch.from_numpy(raw).double().requires_grad_(True)
    outs = _torch_raw2outputs(rt, torch.from_numpy(z).double(), torch.from_numpy(rd).double(), white)
    sum((o * torch.from_numpy(gi).double()).sum() for o, gi in zip(outs, g)).backward()
    rays = np.zeros((R, 11), np.float32)
    rays[:, 3:6] = rd
    d_raw = composite_backward(T(raw), T(z), T(rays), None, white, T(g[0]), T(g[1]), T(g[2]), T(g[4]), T(g[3]))
    assert l2_err(N(d_raw), rt.grad.numpy()) < 2e-4
    # the loss of the training step touches rgb_map only: same thing against the oracle's closed form
    d2 = composite_backward(T(raw), T(z), T(rays), None, white, T(g[0]), None, None)
    assert l2_err(N(d2), O.raw2outputs_backward(raw, z, rd, g[0], white)) < 2e-4


@pytest.mark.parametrize('tag,D,W', [('small', 4, 64), ('full', 8, 256)])
def test_training_step_gradients(golden, tag, D, W):
    from nerfail_amd import run_nerf as RN
    g = golden('g7_train_grads')
    sc, coarse = hip_nerf(D, W, 31, requires_grad=True)
    sf, fine = hip_nerf(D, W, 32, requires_grad=True)
    rays, target = g[tag + '_rays'], g[tag + '_target']
    r = RN.render_rays(T(rays), coarse, None, 64, retraw=True, N_importance=128, network_fine=fine, white_bkgd=True,
                       perturb=1., t_rand=T(g[tag + '_t_rand']), u=T(g[tag + '_u']))
    loss = RN.img2mse(r['rgb_map'], T(target)) + RN.img2mse(r['rgb0'], T(target))
    loss.backward()
    assert abs(float(loss.detach()) - float(g[tag + '_loss'])) < 1e-5 * abs(float(g[tag + '_loss']))
    assert rel_err(N(r['rgb_map']), g[tag + '_rgb_map']) < 1e-4
    # Per-parameter bound from a MEASURED quantity (as :test_mlp_backward_on_identical_inputs does for the
    # isolated MLP): fixture g7 holds, for this very step, the reference's own gradients in fp32 AND in fp64 (same rays, draws
    # and weights) - their L2 distance per parameter is what rounding alone does to the reference (ReLU kinks and
    # importance-sampling bins that flip included: 1e-7 .. 5e-3 depending on the layer). The HIP step must agree with the
    # reference's fp32 gradient within 2 x that + eps, eps = 2e-6 (sin/cos, MFMA summation order) for EVERY parameter.
    # (Rounds 3-4 granted alpha_linear.* 3e-4: the composite backward's suffix sums sum_{k>i} g_k w_k were scanned in fp32 where
    # torch's CPU cumsum accumulates in double. Round 5: composite_bwd.hip scans in double; the exception is gone.)
    ref = O.train_step_grads(rays, sc, sf, target, t_rand=g[tag + '_t_rand'], u=g[tag + '_u'], D=D, W=W)
    worst, lines = 0.0, []
    for nm, net in (('coarse', coarse), ('fine', fine)):
        for k, p in net.named_parameters():
            got = N(p.grad)
            e = l2_err(got, g['%s_%s_grad_%s' % (tag, nm, k)])                      # vs the reference's autograd (fp32), whole tensor
            spread = float(g['%s_%s_referr_%s' % (tag, nm, k)])
            bound = 2 * spread + 2e-6
            worst = max(worst, e / bound)
            lines.append('%-6s %-26s err %.2e  reference fp32-vs-fp64 %.2e  bound %.2e' % (nm, k, e, spread, bound))
            assert l2_err(got, ref['grads_' + nm][k]) < 5e-3, (nm, k)              # and the float64-backward oracle, loosely
    print('\n'.join(lines))
    print('HIP training step (%s) vs the reference step: worst error / bound = %.2f' % (tag, worst))
    assert worst <= 1.0, '\n'.join(lines)


def test_training_step_gradients_headline_shape_on_trained_weights(golden):
    """VERDICT r5 item 1, training half: g22 holds ONE training step (RN:776-791) of the REFERENCE on the trained D=8 W=256 pair
    - 1 024 rays, perturbed draws, sphere colours as target - with its autograd in fp32 and in fp64. On a trained (sharp)
    density the reference's own two precisions differ by up to 2e-3 in a fine-network parameter's gradient (importance bins
    that flip move whole samples); per parameter that measured spread is the bound: HIP within 2 x spread + 2e-6 of the
    reference's fp32 gradient, EVERY parameter of both networks."""
    from conftest import trained_pair_inputs
    from nerfail_amd import run_nerf as RN
    from nerfail_amd.run_nerf_helpers import NeRF
    g = golden('g22_trained_pair_d8')
    sc, sf, rays, t_rand, u = trained_pair_inputs(g)
    tr = g['train_pick']

    def net(sd):
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.requires_grad_(True).to(dev())
    coarse, fine = net(sc), net(sf)
    target = T(g['train_target'])
    r = RN.render_rays(T(rays[tr]), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                       t_rand=T(t_rand[tr]), u=T(u[tr]))
    loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
    loss.backward()
    assert abs(float(loss.detach()) - float(g['train_loss'])) < 1e-4 * abs(float(g['train_loss']))
    d = np.abs(N(r['rgb_map']) - g['train_rgb_map']).max(1)
    print('HIP training forward on g22: %d of %d rays beyond 1e-4 of the reference (median %.1e)' % (int((d > 1e-4).sum()), d.size, np.median(d)))
    assert (d > 1e-4).sum() <= 8 and np.median(d) < 1e-6
    worst, lines = 0.0, []
    for nm, n_ in (('coarse', coarse), ('fine', fine)):
        for k, p in n_.named_parameters():
            e = l2_err(N(p.grad), g['train_%s_grad_%s' % (nm, k)])
            spread = float(g['train_%s_referr_%s' % (nm, k)])
            bound = 2 * spread + 2e-6
            worst = max(worst, e / bound)
            lines.append('%-6s %-26s err %.2e  reference fp32-vs-fp64 %.2e  bound %.2e' % (nm, k, e, spread, bound))
    print('\n'.join(lines))
    print('HIP training step on the trained D8 W256 pair vs the reference step: worst error / bound = %.2f' % worst)
    assert worst <= 1.0, '\n'.join(lines)


def test_training_loop_adam_reduces_loss():
    """RN:776-801 shape of the loop: render -> mse -> backward -> Adam.step; weights change in place, so the packed
    images must be rebuilt every step (cache keyed on parameter versions)."""
    from nerfail_amd import run_nerf as RN
    _, coarse = hip_nerf(4, 64, 41, requires_grad=True)
    _, fine = hip_nerf(4, 64, 42, requires_grad=True)
    from nerfail_amd.optim import Adam
    opt = Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, betas=(0.9, 0.999))
    rays = T(synth.ray_batch(256, seed=7))
    target = T(np.random.RandomState(0).uniform(size=(256, 3)).astype(np.float32))
    gen = torch.Generator(device=dev()).manual_seed(0)
    losses = []
    for it in range(12):
        t_rand = torch.rand((256, 64), device=dev(), generator=gen)
        u = torch.rand((256, 128), device=dev(), generator=gen)
        r = RN.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                           t_rand=t_rand, u=u)
        loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] and np.isfinite(losses).all()


def test_single_pass_and_shared_network_gradients():
    """N_importance = 0 (cfg1) and network_fine=None (fine pass reuses network_fn, RN:399): grads still match the oracle."""
    from nerfail_amd import run_nerf as RN
    sd, net = hip_nerf(4, 64, 51, requires_grad=True)
    rays = synth.ray_batch(40, seed=3)
    target = np.random.RandomState(1).uniform(size=(40, 3)).astype(np.float32)
    r = RN.render_rays(T(rays), net, None, 64, white_bkgd=True)
    RN.img2mse(r['rgb_map'], T(target)).backward()
    z0 = O.coarse_z_vals(rays[:, 6:7], rays[:, 7:8], 64)
    pts0 = rays[:, None, 0:3] + rays[:, None, 3:6] * z0[:, :, None]
    raw0, cache0 = O._mlp_forward_cached(sd, pts0.astype(np.float32), rays[:, -3:], 4, 64)
    rgb0 = O.raw2outputs(raw0, z0, rays[:, 3:6], None, True)[0]
    d0 = O.raw2outputs_backward(raw0, z0, rays[:, 3:6], 2.0 * (rgb0 - target) / rgb0.size, True)
    ref = O.mlp_backward(sd, cache0, d0, 4, 64)
    for k, p in net.named_parameters():
        assert l2_err(N(p.grad), ref[k]) < 5e-3, k


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
@pytest.mark.parametrize('R', [1, 5, 33])
def test_odd_ray_counts_gradients(precision, R):
    """Tile counts that are not a multiple of the 4 waves of a workgroup (R = 33: 66 coarse / 198 fine tiles) and tiny
    batches (R = 1: 2 / 6 tiles, fewer than one workgroup per layer group): partially filled workgroups, dead waves and
    near-empty work partitions of every training kernel, in both precisions, against the oracle. With 1 or 5 rays the
    scalar gradients (alpha bias: a cancelling sum over a few hundred samples) inherit the fine pass's sensitivity to
    one-ulp differences in z_samples (fine `raw` agrees with the reference to 1e-2 only, tests/test_hip_nerf.py), so
    the bound is 2e-2 there and the usual 5e-3 at 33 rays."""
    from nerfail_amd import run_nerf as RN
    sc, coarse = hip_nerf(4, 64, 71, requires_grad=True, precision=precision)
    sf, fine = hip_nerf(4, 64, 72, requires_grad=True, precision=precision)
    rs = np.random.RandomState(R)
    rays = synth.ray_batch(R, seed=70 + R)
    target = rs.uniform(size=(R, 3)).astype(np.float32)
    t_rand, u = rs.uniform(size=(R, 64)).astype(np.float32), rs.uniform(size=(R, 128)).astype(np.float32)
    ref = O.train_step_grads(rays, sc, sf, target, t_rand=t_rand, u=u, D=4, W=64)
    r = RN.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                       t_rand=T(t_rand), u=T(u))
    loss = RN.img2mse(r['rgb_map'], T(target)) + RN.img2mse(r['rgb0'], T(target))
    loss.backward()
    assert abs(float(loss.detach()) - ref['loss']) < 1e-5 * abs(ref['loss'])
    bound = 5e-3 if R >= 33 else 2e-2
    for tag, net in (('grads_coarse', coarse), ('grads_fine', fine)):
        for k, p in net.named_parameters():
            assert l2_err(N(p.grad), ref[tag][k]) < bound, (tag, k)


@pytest.mark.parametrize('dw_kernel', ['reg', 'lds'])
def test_full_width_odd_tiles_weight_gradients(dw_kernel, monkeypatch):
    """D=8 W=256 (the only width with the LDS-staged weight-gradient kernel) on 33 rays: both weight-gradient kernels,
    exact-f32, against each other's reference - the oracle - through the norm of every gradient and 256 entries."""
    from nerfail_amd import run_nerf as RN
    monkeypatch.setenv('NERFAIL_DW_KERNEL', dw_kernel)
    sc, coarse = hip_nerf(8, 256, 81, requires_grad=True)
    sf, fine = hip_nerf(8, 256, 82, requires_grad=True)
    R = 33
    rs = np.random.RandomState(8)
    rays = synth.ray_batch(R, seed=83)
    target = rs.uniform(size=(R, 3)).astype(np.float32)
    t_rand, u = rs.uniform(size=(R, 64)).astype(np.float32), rs.uniform(size=(R, 128)).astype(np.float32)
    ref = O.train_step_grads(rays, sc, sf, target, t_rand=t_rand, u=u, D=8, W=256)
    r = RN.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                       t_rand=T(t_rand), u=T(u))
    loss = RN.img2mse(r['rgb_map'], T(target)) + RN.img2mse(r['rgb0'], T(target))
    loss.backward()
    assert abs(float(loss.detach()) - ref['loss']) < 1e-5 * abs(ref['loss'])
    for tag, net in (('grads_coarse', coarse), ('grads_fine', fine)):
        for k, p in net.named_parameters():
            assert l2_err(N(p.grad), ref[tag][k]) < 5e-3, (tag, k)


@pytest.mark.parametrize('dw_kernel', ['lds', 'reg'])
def test_mlp_backward_on_identical_inputs(golden, dw_kernel, monkeypatch):
    """Row a12 isolated (fixture g16): the reference's Embedder + NeRF.forward under autograd on FIXED points, upstream
    d_raw spanning 6 decades -> every parameter gradient of D=8 W=256. The bound is not a flat tolerance: per
    parameter, 2 x the reference's own fp32-vs-fp64 L2 error (stored in the fixture, 0.5-9e-7) plus 2e-6 for what
    differs by construction (the kernel's own sin/cos and MFMA / atomic summation order over 2 048 samples)."""
    monkeypatch.setenv('NERFAIL_DW_KERNEL', dw_kernel)
    from hiputil import hip_mlp_grads
    g = golden('g16_mlp_backward')
    sd, net = hip_nerf(8, 256, int(g['seed']), requires_grad=True)
    grads = hip_mlp_grads(net, T(g['pts']), T(g['dirs']), T(g['d_raw']))
    worst, lines = 0.0, []
    for k in sd:
        e, ref_e = l2_err(grads[k], g['grad_' + k]), float(g['ref_err_' + k])
        bound = 2 * ref_e + 2e-6
        worst = max(worst, e / bound)
        lines.append('%-26s err %.2e  reference fp32-vs-fp64 %.2e  bound %.2e' % (k, e, ref_e, bound))
    print('\n'.join(lines))
    print('HIP f32 backward (%s) vs reference autograd on identical inputs: worst error / bound = %.2f' % (dw_kernel, worst))
    assert worst <= 1.0, '\n'.join(lines)


def _step_grads(coarse, fine, rays, target, t_rand, u):
    from nerfail_amd import run_nerf as RN
    for p in list(coarse.parameters()) + list(fine.parameters()):
        p.grad = None
    r = RN.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                       t_rand=T(t_rand), u=T(u))
    loss = RN.img2mse(r['rgb_map'], T(target)) + RN.img2mse(r['rgb0'], T(target))
    loss.backward()
    return {('c' if n is coarse else 'f') + k: p.grad.clone() for n in (coarse, fine) for k, p in n.named_parameters()}


def test_training_step_is_bitwise_reproducible(golden):
    """the W = 256 weight gradients carry no float atomics (per-workgroup partial slabs, summed in
    workgroup order), so two runs of the same step give the SAME BITS in every parameter gradient."""
    g = golden('g7_train_grads')
    _, coarse = hip_nerf(8, 256, 31, requires_grad=True)
    _, fine = hip_nerf(8, 256, 32, requires_grad=True)
    args = (g['full_rays'], g['full_target'], g['full_t_rand'], g['full_u'])
    a = _step_grads(coarse, fine, *args)
    for rep in range(3):
        b = _step_grads(coarse, fine, *args)
        for k in a:
            assert torch.equal(a[k], b[k]), (rep, k)


def test_joint_backward_equals_separate_launches(golden):
    """The coarse and the fine network's backward run as ONE launch per kernel (nerfail_mlp_bwd_data2, two-network
    nerfail_mlp_bwd_weights): same gradients as one launch per network (different partition -> last-bit differences
    only), and a network evaluated for both passes (network_fine=None) accumulates both."""
    from nerfail_amd import _train, _lib
    rs = np.random.RandomState(5)
    _, n0 = hip_nerf(8, 256, 41, requires_grad=True)
    _, n1 = hip_nerf(8, 256, 42, requires_grad=True)
    R, Nc, Nf = 96, 64, 192
    M0, M1 = R * Nc, R * Nf
    pts = T(rs.uniform(-2, 2, (R, Nc + Nf, 3)).astype(np.float32))
    vd = T(synth.ray_batch(R, seed=3)[:, 8:11].copy())
    nA = _train.acts_floats(n0, M0)
    acts = torch.empty((nA + _train.acts_floats(n1, M1),), device=dev())
    _train.mlp_fwd_train(n0, pts[:, :Nc].contiguous(), vd, acts=acts[:nA])
    _train.mlp_fwd_train(n1, pts[:, Nc:].contiguous(), vd, acts=acts[nA:])
    d_raw = T((rs.normal(size=(M0 + M1, 4)) * 10.0 ** rs.uniform(-5, 0, (M0 + M1, 1))).astype(np.float32))
    gj0, gj1 = _train._new_grads(n0, False), _train._new_grads(n1, False)
    _train.mlp_backward2(n0, d_raw, acts, gj0, M0, n1, gj1, M1)
    gs0, gs1 = _train._new_grads(n0, False), _train._new_grads(n1, False)
    _train.mlp_backward2(n0, d_raw[:M0], acts[:nA], gs0, M0, None, None, 0)
    _train.mlp_backward2(n1, d_raw[M0:], acts[nA:], gs1, M1, None, None, 0)
    for a, b in zip(gj0 + gj1, gs0 + gs1):
        assert l2_err(N(a), N(b)) < 2e-6
    # accumulate flag: += on top of what is there
    ga = [t.clone() for t in gs0]
    _train.mlp_backward2(n0, d_raw[:M0], acts[:nA], ga, M0, None, None, 0, accumulate=True)
    for a, b in zip(ga, gs0):
        assert l2_err(N(a), 2.0 * N(b)) < 2e-6


@pytest.mark.parametrize('R0,R1,N0', [(64, 0, 64), (20, 37, 64), (3, 1, 64), (128, 384, 64),
                                      (5, 0, 37), (1, 0, 31), (33, 0, 97),          # M % 32 != 0 - the `sraw < a.M` padding path
                                      (600, 0, 64), (300, 300, 64)])                # more tiles than 4 x 256 per network: several rounds
def test_lds_ring_backward_data_stores_the_same_bits_as_the_register_kernel(R0, R1, N0):
    """Round 5: nerfail_mlp_bwd_data2 runs nerf_mlp_bwd_data_lds_kernel at W = 256 (the transposed image through the LDS weight
    ring, dZ formed lazily from the ReLU bits where it is consumed and stored from there, one network per workgroup). Every
    stored dZ - all layers, both networks of a joint launch, ragged last tiles, fewer tiles than waves - must be BITWISE what the
    register-streamed nerf_mlp_bwd_data_kernel stores (same products, same order of additions per accumulator)."""
    from nerfail_amd import _lib, _train
    lib = _lib.load()
    _, n0 = hip_nerf(8, 256, 61, requires_grad=True)
    _, n1 = hip_nerf(8, 256, 62, requires_grad=True)
    rs = np.random.RandomState(R0 + R1)
    N1 = 192
    M0, M1 = R0 * N0, R1 * N1
    M0 = (M0 + 31) // 32 * 32 if M1 else M0                    # (a joint launch needs whole tiles of the first network)
    vd = torch.nn.functional.normalize(T(rs.normal(size=(max(R0, R1, 1), 3)).astype(np.float32)), dim=-1)
    acts = torch.empty((_train.acts_floats(n0, M0) + (_train.acts_floats(n1, M1) if M1 else 0),), device=dev())
    nA = _train.acts_floats(n0, M0)
    R0e = M0 // N0
    assert R0e * N0 == M0 and (M1 > 0 or N0 == 64 or M0 % 32 != 0)
    _train.mlp_fwd_train(n0, T(rs.normal(size=(R0e, N0, 3)).astype(np.float32)), vd[:R0e].contiguous(), acts=acts[:nA])
    if M1:
        _train.mlp_fwd_train(n1, T(rs.normal(size=(R1, N1, 3)).astype(np.float32)), vd[:R1].contiguous(), acts=acts[nA:])
    d_raw = T((rs.normal(size=(M0 + M1, 4)) * 10.0 ** rs.uniform(-4, 0, (M0 + M1, 1))).astype(np.float32))
    (p0, pT0), (p1, pT1) = _train.packed_both(n0), _train.packed_both(n1)
    nz = _train.dz_floats(n0, M0) + (_train.dz_floats(n1, M1) if M1 else 0)
    out = {}
    for which in (1, 2):
        dz = torch.full((nz,), float('nan'), device=dev())
        prev = lib.nerfail_mlp_bwd_select(which)
        try:
            _lib.check(lib.nerfail_mlp_bwd_data2(_lib.dev(p0), _lib.dev(pT0), M0, _lib.dev(p1) if M1 else None, _lib.dev(pT1) if M1 else None,
                                                 M1, 8, 256, n0._skip(), _lib.dev(d_raw), _lib.dev(acts), _lib.dev(dz), _lib.stream()))
        finally:
            lib.nerfail_mlp_bwd_select(prev)
        torch.cuda.synchronize()
        out[which] = dz
    assert not torch.isnan(out[1]).any()
    assert torch.equal(out[1], out[2])
