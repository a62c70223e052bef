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
    worst = {}
    for k in ('rgb_map', 'acc_map', 'rgb0'):
        worst[k] = rel_err(N(outs['f16x3'][k]), N(outs['f32'][k]))
        d = np.abs(N(outs['f16x3'][k]) - N(outs['f32'][k])).reshape(4096, -1).max(1)
        worst[k + '_rays_over_1e-4'] = int((d > 1e-4).sum())
    sd_c = {k: N(v) for k, v in coarse.state_dict().items()}
    sd_f = {k: N(v) for k, v in fine.state_dict().items()}
    nref = 256
    ref = O.render_rays(N(rays[:nref]), sd_c, 64, 128, sd_f, white_bkgd=True)
    for prec in ('f32', 'f16x3'):
        for k in ('rgb_map', 'acc_map', 'rgb0'):
            worst['%s_vs_oracle_%s' % (prec, k)] = rel_err(N(outs[prec][k][:nref]), ref[k])
    for prec in ('f32', 'f16x3'):
        worst['%s_vs_oracle_acc_map_abs' % prec] = float(np.abs(N(outs[prec]['acc_map'][:nref]) - ref['acc_map']).max())
    worst['acc_map_abs'] = float(np.abs(N(outs['f16x3']['acc_map']) - N(outs['f32']['acc_map'])).max())
    print('on trained weights: %s' % {k: ('%.1e' % v if isinstance(v, float) else v) for k, v in worst.items()})
    TRAINED_WORST.update(worst)
    # The exact-f32 kernel on trained weights: the 1e-4 bound of every other parity test, against the oracle.
    for k in ('f32_vs_oracle_rgb_map', 'f32_vs_oracle_rgb0'):
        assert worst[k] < 1e-4, (k, worst[k])
    # The split kernel, coarse pass (no resampling in between): same bound, vs the exact kernel and vs the oracle.
    for k in ('rgb0', 'f16x3_vs_oracle_rgb0'):
        assert worst[k] < 1e-4, (k, worst[k])
    # The split kernel through the WHOLE path. MEASURED in round 4 on two trainings that differ in the last bits of one kernel:
    #   run A: rgb_map 6.4e-5, acc_map 6.4e-5 absolute, 0 of 4096 rays beyond 1e-4;
    #   run B: rgb_map 4.1e-4, 2 of 4096 rays beyond 1e-4.
    # A trained density is sharp (sigma * delta ~ 10): the split products' ~1e-6 relative error of raw sigma is ~50x larger in
    # the coarse weights than on random-init nets (3.6e-7 there), and a coarse weight that moves by 1e-5 can move an
    # importance sample across a bin (RH:226-240), i.e. one of the 128 fine samples of that ray by a whole bin. The exact
    # kernel has the same discontinuity, 50x more rarely. So: NOT within 1e-4 on every ray of a trained scene - after only
    # 2 000 steps. This is the evidence on which f16x3 STAYS OPT-IN (DESIGN.md, K3+K4 split-precision variant); what is
    # asserted here is that it stays a rare, bounded event: >= 99.5 % of the rays inside 1e-4, none beyond 5e-3.
    assert worst['rgb_map_rays_over_1e-4'] <= 20 and worst['acc_map_rays_over_1e-4'] <= 20, worst
    assert worst['rgb_map'] < 5e-3 and worst['acc_map_abs'] < 5e-3, worst
    # accumulated opacity of the exact kernel (in [0, 1]): absolute - a trained scene has nearly EMPTY rays (acc ~ 1e-2 and
    # less), where conftest.rel_err (floor 1e-3 of the maximum) turns an absolute 5e-6 into "2e-4 relative".
    assert worst['f32_vs_oracle_acc_map_abs'] < 2e-5, worst
