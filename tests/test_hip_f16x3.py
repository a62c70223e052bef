"""-m gpu: the split-precision ("f16x3") MLP kernel against the SAME golden vectors and the SAME 1e-4 bound as the
exact-f32 kernel, plus a direct error comparison of the two kernels against a float64 evaluation of the network."""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_err
from hiputil import T, N, hip_nerf
from oracle import nerf as O

pytestmark = pytest.mark.gpu


def _run(net, pts, dirs):
    from nerfail_amd.run_nerf import _mlp_points
    return N(_mlp_points(net, T(pts), T(dirs)))


@pytest.mark.parametrize('D,W', [(8, 256), (4, 64)])
def test_f16x3_mlp_matches_reference(golden, D, W):
    g = golden('g3_nerf_forward')
    seed = int(g['seed_D%dW%d' % (D, W)])
    _, net = hip_nerf(D, W, seed, precision='f16x3')
    raw = _run(net, g['pts'].reshape(8, 64, 3), g['dirs'][:8])
    if (D, W) == (8, 256):
        assert rel_err(raw, g['run_network_raw']) < 1e-4
    sd = synth.nerf_state_dict(D=D, W=W, seed=seed)
    ref = O.run_network(sd, g['pts'].reshape(8, 64, 3), g['dirs'][:8], D=D, W=W)
    assert rel_err(raw, ref) < 1e-4
    # ragged sizes
    for r, n in ((1, 64), (3, 21), (2, 33)):
        got = _run(net, g['pts'][:r * n].reshape(r, n, 3), g['dirs'][:r])
        assert rel_err(got, O.run_network(sd, g['pts'][:r * n].reshape(r, n, 3), g['dirs'][:r], D=D, W=W)) < 1e-4


def test_f16x3_error_is_fp32_level():
    """Both kernels vs the network evaluated in float64: the split-precision error must stay within a small factor
    of the exact-f32 kernel's own rounding error (it is not a reduced-precision path)."""
    rs = np.random.RandomState(0)
    R, Ns = 64, 192
    pts = rs.uniform(-2, 2, (R, Ns, 3)).astype(np.float32)
    dirs = rs.normal(size=(R, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    sd, n32 = hip_nerf(8, 256, 5)
    _, n16 = hip_nerf(8, 256, 5, precision='f16x3')
    sd64 = {k: v.astype(np.float64) for k, v in sd.items()}
    flat = pts.reshape(-1, 3)
    d = np.broadcast_to(dirs[:, None, :], pts.shape).reshape(-1, 3)
    emb = np.concatenate([O.embed(flat, 10), O.embed(d, 4)], -1).astype(np.float64)   # same float32 encoding values
    ref = O.nerf_forward.__wrapped__(sd64, emb) if hasattr(O.nerf_forward, '__wrapped__') else None
    # float64 forward (O.nerf_forward casts to float32, so restate the few lines here in float64)
    h = emb[:, :63]
    for i in range(8):
        h = np.maximum(h @ sd64['pts_linears.%d.weight' % i].T + sd64['pts_linears.%d.bias' % i], 0)
        if i == 4:
            h = np.concatenate([emb[:, :63], h], -1)
    alpha = h @ sd64['alpha_linear.weight'].T + sd64['alpha_linear.bias']
    feat = h @ sd64['feature_linear.weight'].T + sd64['feature_linear.bias']
    hv = np.maximum(np.concatenate([feat, emb[:, 63:]], -1) @ sd64['views_linears.0.weight'].T + sd64['views_linears.0.bias'], 0)
    ref = np.concatenate([hv @ sd64['rgb_linear.weight'].T + sd64['rgb_linear.bias'], alpha], -1).reshape(R, Ns, 4)
    e32 = np.abs(_run(n32, pts, dirs) - ref).max()
    e16 = np.abs(_run(n16, pts, dirs) - ref).max()
    scale = np.abs(ref).max()
    assert e32 < 2e-5 * scale and e16 < 2e-5 * scale, (e32, e16, scale)
    assert e16 < 8 * e32 + 1e-7 * scale, (e32, e16)


def test_f16x3_render_matches_reference(golden):
    from nerfail_amd import nerf_to_coord as NC
    g = golden('g6_render_rays')
    _, coarse = hip_nerf(8, 256, int(g['cfg2_seed_coarse']), precision='f16x3')
    _, fine = hip_nerf(8, 256, int(g['cfg2_seed_fine']), precision='f16x3')
    r = NC.render_rays(T(g['cfg2_rays']), coarse, None, 64, retraw=True, N_importance=128, network_fine=fine, white_bkgd=True)
    for k in ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'pts_max'):
        assert rel_err(N(r[k]), g['cfg2_det_' + k]) < 1e-4, k
    r = NC.render_rays(T(g['cfg2_rays']), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True,
                       perturb=1., t_rand=T(g['cfg2_t_rand']), u=T(g['cfg2_u']))
    for k in ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'z_std', 'pts_max'):
        assert rel_err(N(r[k]), g['cfg2_pert_' + k]) < 1e-4, k


@pytest.mark.parametrize('tag,D,W', [('small', 4, 64), ('full', 8, 256)])
def test_f16x3_training_forward_gradients(golden, tag, D, W):
    """Training step on the split-precision kernels (f16x3 forward-with-activations and backward-data, bf16x3
    weight gradients): loss and parameter gradients vs the reference's autograd, same bounds as
    tests/test_hip_train.py."""
    from conftest import l2_err
    from nerfail_amd import run_nerf as RN
    g = golden('g7_train_grads')
    _, coarse = hip_nerf(D, W, 31, requires_grad=True, precision='f16x3')
    _, fine = hip_nerf(D, W, 32, requires_grad=True, precision='f16x3')
    r = RN.render_rays(T(g[tag + '_rays']), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True,
                       perturb=1., t_rand=T(g[tag + '_t_rand']), u=T(g[tag + '_u']))
    target = T(g[tag + '_target'])
    loss = RN.img2mse(r['rgb_map'], target) + RN.img2mse(r['rgb0'], target)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[tag + '_loss'])) < 1e-5 * abs(float(g[tag + '_loss']))
    for nm, net in (('coarse', coarse), ('fine', fine)):
        for k, p in net.named_parameters():
            got = N(p.grad)
            if tag == 'small':
                assert l2_err(got, g['small_%s_grad_%s' % (nm, k)]) < 5e-3, (nm, k)
            else:
                refn = float(g['full_%s_gradnorm_%s' % (nm, k)])
                assert abs(np.linalg.norm(got.astype(np.float64)) - refn) < 5e-3 * refn, (nm, k)


@pytest.mark.parametrize('dw_kernel', ['reg', 'lds'])
def test_split_backward_kernels_match_f32_kernels(dw_kernel, monkeypatch):
    """Backward-data (f16x3) and weight-gradient (bf16x3) kernels against the exact-f32 kernels on the SAME saved
    activations (same ReLU masks, so no discrete differences): every parameter gradient of a D=8 W=256 network agrees
    to 2e-5 L2 (measured: 1e-6 backward-data, 5e-6 weight gradients) although the upstream gradient spans ~8 orders
    of magnitude between samples (as ray weights do). A single per-wave scale in the f16 split misses this bound by
    two orders of magnitude at the first layers (the fp16 lo halves go subnormal)."""
    from conftest import l2_err
    from hiputil import hip_mlp_grads
    monkeypatch.setenv('NERFAIL_DW_KERNEL', dw_kernel)     # register-fed (default) / LDS-staged bf16x3 weight-gradient kernel
    rng = np.random.default_rng(11)
    R, n = 128, 64
    pts = T(rng.uniform(-1.5, 1.5, (R, n, 3)).astype(np.float32))
    d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    dirs = T(d.astype(np.float32))
    d_raw = T((rng.normal(size=(R, n, 4)) * np.exp(3. * rng.normal(size=(R, n, 1)))).astype(np.float32))
    _, net = hip_nerf(8, 256, 31, requires_grad=True)
    ref = hip_mlp_grads(net, pts, dirs, d_raw, 'f32', 'f32', 'f32')
    for bd, dw in (('split', 'f32'), ('f32', 'split'), ('split', 'split')):
        got = hip_mlp_grads(net, pts, dirs, d_raw, 'f32', bd, dw)
        worst = max((l2_err(got[k], ref[k]), k) for k in ref)
        assert worst[0] < 2e-5, (bd, dw, worst)


def test_split_gradients_vs_float64_truth():
    """All three split kernels together against a float64 torch evaluation of the same network and upstream gradient:
    within the same bound as the exact-f32 kernels (both are limited by ReLU-mask flips of the fp32 forward)."""
    from hiputil import hip_mlp_grads, torch_nerf_mlp
    rng = np.random.default_rng(12)
    R, n = 64, 64
    pts = T(rng.uniform(-1.5, 1.5, (R, n, 3)).astype(np.float32))
    d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    dirs = T(d.astype(np.float32))
    d_raw = T((rng.normal(size=(R, n, 4)) * np.exp(2. * rng.normal(size=(R, n, 1)))).astype(np.float32))
    sd, net = hip_nerf(8, 256, 31, requires_grad=True)
    raw, P = torch_nerf_mlp(sd, pts.reshape(-1, 3), dirs[:, None, :].expand(R, n, 3).reshape(-1, 3), torch.float64)
    (raw * d_raw.reshape(-1, 4).double()).sum().backward()
    truth = {k: v.grad.cpu().numpy() for k, v in P.items()}
    e = {}
    for tag, modes in (('f32', ('f32', 'f32', 'f32')), ('split', ('split', 'split', 'split'))):
        got = hip_mlp_grads(net, pts, dirs, d_raw, *modes)
        e[tag] = {k: np.linalg.norm(got[k] - truth[k]) / np.linalg.norm(truth[k]) for k in truth}
    for k in truth:
        assert e['f32'][k] < 5e-3 and e['split'][k] < 5e-3, (k, e['f32'][k], e['split'][k])
        assert e['split'][k] < 10 * e['f32'][k] + 3e-4, (k, e['f32'][k], e['split'][k])


def test_f16x3_rejects_weights_outside_its_range():
    """the split-precision image stores fp16(w * 2^10); a weight of magnitude >= 64 would silently become inf.
    The mirror refuses the mode instead (the exact-f32 kernel has no such limit)."""
    from nerfail_amd.run_nerf import _mlp_points
    _, net = hip_nerf(4, 64, 11, precision='f16x3')
    with torch.no_grad():
        net.pts_linears[1].weight[3, 5] = 70.0
    pts = T(np.zeros((4, 8, 3), np.float32))
    vd = T(np.tile(np.array([[0., 0., 1.]], np.float32), (4, 1)))
    with pytest.raises(ValueError, match='f16x3'):
        _mlp_points(net, pts, vd)
    net.precision = 'f32'
    assert torch.isfinite(_mlp_points(net, pts, vd)).all()
    # EVERY later pack is checked too (without a stall: reported by the next call), whatever changed the weights
    _, net = hip_nerf(4, 64, 12, precision='f16x3')
    assert torch.isfinite(_mlp_points(net, pts, vd)).all()                  # first pack: in range
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd['views_linears.0.weight'][2, 7] = -90.0
    net.load_state_dict(sd)
    with pytest.raises(ValueError, match='f16x3'):
        _mlp_points(net, pts, vd)                                           # this pack measures ...
        torch.cuda.synchronize()
        _mlp_points(net, pts, vd)                                           # ... and the next call reports


TRAINED_WORST = {}


def _sphere_target(rays):
    """Analytic scene for a 'trained-like' net: a unit sphere shaded by its normal on a white background (white_bkgd)."""
    o, d = rays[:, 0:3].astype(np.float64), rays[:, 3:6].astype(np.float64)
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    b = (o * dn).sum(1)
    disc = b * b - ((o * o).sum(1) - 1.0)
    t = -b - np.sqrt(np.maximum(disc, 0.0))
    hit = (disc > 0) & (t > 0)
    nrm = o + dn * t[:, None]
    rgb = np.where(hit[:, None], 0.5 + 0.5 * nrm, 1.0)
    return rgb.astype(np.float32), hit


def test_f16x3_on_trained_like_weights():
    """everything above judges f16x3 on seeded random-init weights. Here both networks are TRAINED first -
    2 000 Adam steps of the product's own exact-f32 training step (RN:776-801: 1024 random rays of 40 poses around an analytic
    sphere, lr 5e-4 with the reference's decay) - and then asked the two questions that decide whether the mode could ever
    be a default: (i) how far are the weights from the hard |w| < 64 limit of the fp16 image, (ii) does render_rays on the
    trained weights still agree with the exact-f32 kernel and with the float64-accumulating oracle at the 1e-4 bound."""
    from nerfail_amd import run_nerf as RN
    from nerfail_amd.optim import Adam
    from nerfail_amd.run_nerf import ray_gen
    from hiputil import dev
    sc, coarse = hip_nerf(8, 256, 71, requires_grad=True)
    sf, fine = hip_nerf(8, 256, 72, requires_grad=True)
    params = list(coarse.parameters()) + list(fine.parameters())
    opt = Adam(params, lr=5e-4, betas=(0.9, 0.999))
    Hs = Ws = 100
    focal, K = synth.lego_intrinsics(Hs, Ws)
    rays_all = torch.cat([ray_gen(Hs, Ws, K, synth.pose_spherical(float(th), -30., 4.)[:3, :4], 2., 6.) for th in np.linspace(-180, 180, 41)[:-1]])
    tgt_np, hit = _sphere_target(N(rays_all))
    assert 0.05 < hit.mean() < 0.6
    tgt_all = T(tgt_np)
    gen = torch.Generator(device=dev()).manual_seed(0)
    steps, first, last = 2000, None, None
    for it in range(steps):
        sel = torch.randint(0, rays_all.shape[0], (1024,), device=dev(), generator=gen)
        r = RN.render_rays(rays_all[sel].contiguous(), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                           t_rand=torch.rand((1024, 64), device=dev(), generator=gen), u=torch.rand((1024, 128), device=dev(), generator=gen))
        loss = RN.img2mse(r['rgb_map'], tgt_all[sel]) + RN.img2mse(r['rgb0'], tgt_all[sel])
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:                                   # RN:796-800 (lrate_decay = 500 -> 500 000 steps)
            g['lr'] = 5e-4 * 0.1 ** ((it + 1) / 500000.)
        if it == 0:
            first = float(loss.detach())
    last = float(loss.detach())
    assert np.isfinite(last) and last < 0.5 * first, (first, last)           # it really learned the scene
    wmax0 = max(float(np.abs(v).max()) for k, v in {**sc, **sf}.items() if k.endswith('weight'))
    wmax = max(float(p.detach().abs().max()) for n in (coarse, fine) for k, p in n.named_parameters() if k.endswith('weight'))
    margin = coarse.F16X3_MAX_WEIGHT / wmax
    print('trained-like weights: loss %.4f -> %.4f in %d steps; max |w| %.3f at init, %.3f trained: %.0fx below the f16x3 limit of %g'
          % (first, last, steps, wmax0, wmax, margin, coarse.F16X3_MAX_WEIGHT))
    assert margin > 8.0
    # (ii) the same rays through both kernels and through the oracle, on the TRAINED weights
    for n in (coarse, fine):
        n.requires_grad_(False)
    rays = rays_all[torch.arange(0, rays_all.shape[0], rays_all.shape[0] // 4096, device=dev())[:4096]].contiguous()
    outs = {}
    for prec in ('f32', 'f16x3'):
        coarse.precision = fine.precision = prec
        with torch.no_grad():
            outs[prec] = RN.render_rays(rays, coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=0.)
    nref = 1024
    sd_c = {k: N(v) for k, v in coarse.state_dict().items()}
    sd_f = {k: N(v) for k, v in fine.state_dict().items()}
    ref = O.render_rays(N(rays[:nref]), sd_c, 64, 128, sd_f, white_bkgd=True)
    pairs = {'f16x3_vs_f32': (lambda k: N(outs['f16x3'][k]), lambda k: N(outs['f32'][k])),
             'f32_vs_oracle': (lambda k: N(outs['f32'][k][:nref]), lambda k: ref[k]),
             'f16x3_vs_oracle': (lambda k: N(outs['f16x3'][k][:nref]), lambda k: ref[k])}
    worst = {}
    for name, (fa, fb) in pairs.items():
        worst[name + '_rgb0'] = rel_err(fa('rgb0'), fb('rgb0'))                     # coarse pass: no resampling in between
        d = np.abs(fa('rgb_map') - fb('rgb_map')).max(1)
        da = np.abs(fa('acc_map') - fb('acc_map'))
        worst[name + '_rays'] = int(d.size)
        worst[name + '_rays_over_1e-4'] = int(((d > 1e-4) | (da > 1e-4)).sum())
        worst[name + '_median_abs'] = float(np.median(d))
        worst[name + '_max_abs'] = float(max(d.max(), da.max()))
    print('on trained weights: %s' % {k: ('%.1e' % v if isinstance(v, float) else v) for k, v in worst.items()})
    TRAINED_WORST.update(worst)
    # (a) The coarse pass - everything up to the first compositing, no resampling - holds the 1e-4 bound of every other parity
    # test on trained weights too, for the split kernel as for the exact one (measured 4e-6 .. 8e-6).
    for name in pairs:
        assert worst[name + '_rgb0'] < 1e-4, (name, worst)
    # (b) Through the WHOLE path NO pair of implementations agrees within 1e-4 on every ray of a trained scene - not the
    # split kernel with the exact kernel, and not the exact kernel with the numpy oracle. MEASURED in round 4 on three
    # trainings that differ in the last bits of one kernel (4 096 rays split-vs-exact, 256-1 024 rays vs the oracle):
    #   split vs exact: 0, 2 and 2 rays beyond 1e-4 (worst 6e-5, 4e-4, 2e-2); exact vs oracle: 0, 0 and >= 1 ray (worst 1.6e-4
    #   rgb, 3e-4 acc). A trained density is sharp (sigma * delta ~ 10): a coarse weight that moves in its 6th digit can move
    #   an importance sample across a bin (RH:226-240) and with it one of the ray's 128 fine samples by a whole bin. The median
    #   ray agrees to 1e-7. That is a property of hierarchical sampling on sharp densities, not of a kernel; what a kernel can be
    #   held to is that such rays stay RARE: <= 1 % of the rays beyond 1e-4 for every pair.
    # For f16x3 this means: its arithmetic error (coarse pass, median ray) is at the exact kernel's level on trained weights,
    # its weights are 150x inside the fp16 range - and it flips bins about as rarely as the exact kernel does against the
    # oracle. It stays opt-in because "as good as the exact kernel" cannot be asserted ray by ray.
    for name in pairs:
        assert worst[name + '_rays_over_1e-4'] <= 0.01 * worst[name + '_rays'], (name, worst)
        assert worst[name + '_median_abs'] < 1e-5, (name, worst)
