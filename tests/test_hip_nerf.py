"""-m gpu: the HIP render path (through the C ABI) against the golden vectors of the reference and the oracle.

Tolerance: 1e-4 relative float32 (BASELINE.json north_star) measured by conftest.rel_err; tighter where the
kernel is elementwise. Per-sample `raw` / `weights` carry reference-inherent sensitivities documented in
tests/test_oracle_nerf.py and get the same treatment here.
"""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_err, plain_rel_err, trained_pair_inputs, check_against_trained_reference
from hiputil import T, N, hip_nerf, dev
from oracle import nerf as O

pytestmark = pytest.mark.gpu


def test_device_is_gfx950():
    from nerfail_amd import _lib
    assert 'gfx950' in _lib.device_name()


def test_get_rays_and_ray_gen(golden):
    from nerfail_amd.run_nerf_helpers import get_rays
    from nerfail_amd.run_nerf import ray_gen, _pack_rays
    g = golden('g1_get_rays')
    ro, rd = get_rays(16, 16, g['K16'], g['c2w16'])
    assert np.array_equal(N(ro), g['rays_o16'])
    assert rel_err(N(rd), g['rays_d16']) < 1e-6
    ro, rd = get_rays(800, 800, g['K800'], g['c2w800'])
    assert rel_err(N(rd)[g['jj800'], g['ii800']], g['rays_d800']) < 1e-6
    # bit-exact against the oracle's operation order, non-square image
    H, W = 37, 53
    focal, K = synth.lego_intrinsics(H, W)
    c2w = synth.pose_spherical(40., -30., 4.)[:3, :4]
    oo, od = O.get_rays(H, W, K, c2w)
    ro, rd = get_rays(H, W, K, c2w)
    assert np.array_equal(N(ro), oo) and np.array_equal(N(rd), od)
    # packed rays of a pixel range == pack(get_rays)[range] (the multi-GPU shard unit), and == oracle packing
    full = N(_pack_rays(ro, rd, 2., 6.))
    assert rel_err(full, O.pack_rays(oo, od, 2., 6.)) < 1e-6
    part = N(ray_gen(H, W, K, c2w, 2., 6., pix_begin=101, pix_count=777))
    assert np.array_equal(part, full[101:101 + 777])
    assert ray_gen(H, W, K, c2w, 2., 6., pix_begin=5, pix_count=0).shape == (0, 11)


def test_embed(golden):
    from nerfail_amd.run_nerf_helpers import get_embedder
    g = golden('g2_embed')
    e10, d10 = get_embedder(10, 0)
    e4, d4 = get_embedder(4, 0)
    assert (d10, d4) == (63, 27)
    assert np.abs(N(e10(T(g['pts']))) - g['emb_pts']).max() < 1e-6
    assert np.abs(N(e4(T(g['dirs']))) - g['emb_dirs']).max() < 1e-6


@pytest.mark.parametrize('D,W', [(8, 256), (4, 64)])
def test_nerf_forward_embedded(golden, D, W):
    g = golden('g3_nerf_forward')
    sd, net = hip_nerf(D, W, int(g['seed_D%dW%d' % (D, W)]))
    emb = np.concatenate([O.embed(g['pts'], 10), O.embed(g['dirs'], 4)], -1)
    raw = N(net(T(emb)))
    assert rel_err(raw, g['raw_D%dW%d' % (D, W)]) < 1e-4
    assert rel_err(raw, O.nerf_forward(sd, emb, D=D, W=W)) < 1e-4
    # ragged sizes: 1, 31, 33 samples (tiles are 32 wide)
    for m in (1, 31, 33):
        assert rel_err(N(net(T(emb[:m]))), g['raw_D%dW%d' % (D, W)][:m]) < 1e-4


def test_nerf_state_dict_keys_match_reference():
    sd, net = hip_nerf(8, 256, 3)
    assert set(net.state_dict().keys()) == set(sd.keys())


def test_run_network_fused(golden):
    from nerfail_amd.run_nerf import run_network
    from nerfail_amd.run_nerf_helpers import get_embedder
    g = golden('g3_nerf_forward')
    sd, net = hip_nerf(8, 256, 10)
    e10, _ = get_embedder(10, 0)
    e4, _ = get_embedder(4, 0)
    raw = run_network(T(g['pts'].reshape(8, 64, 3)), T(g['dirs'][:8]), net, e10, e4, netchunk=1024 * 64)
    assert rel_err(N(raw), g['run_network_raw']) < 1e-4
    # weights changed in place -> the packed image must be rebuilt
    with torch.no_grad():
        net.rgb_linear.bias.add_(1.0)
    raw2 = run_network(T(g['pts'].reshape(8, 64, 3)), T(g['dirs'][:8]), net, e10, e4)
    assert np.abs(N(raw2)[..., :3] - (g['run_network_raw'][..., :3] + 1.0)).max() < 1e-4


def test_raw2outputs(golden):
    from nerfail_amd.run_nerf import raw2outputs
    g = golden('g4_raw2outputs')
    for Ns in (64, 192):
        raw, z, rd = g['N%d_raw' % Ns], g['N%d_z' % Ns], g['N%d_rays_d' % Ns]
        for wb in (False, True):
            out = raw2outputs(T(raw), T(z), T(rd), 0, wb)
            for k, v in zip(('rgb', 'disp', 'acc', 'weights', 'depth'), out):
                ref = g['N%d_wb%d_%s' % (Ns, int(wb), k)]
                # alpha = 1-exp(-x) cancels for faint samples: one ulp of expf is ~1e-5 of a tiny acc / weight
                assert rel_err(N(v), ref) < (1e-4 if k == 'weights' else 5e-5), (Ns, wb, k)
        out = raw2outputs(T(raw), T(z), T(rd), 0.5, True, noise=T(g['N%d_noise' % Ns] * np.float32(0.5)))
        for k, v in zip(('rgb', 'disp', 'acc', 'weights', 'depth'), out):
            assert rel_err(N(v), g['N%d_noise_%s' % (Ns, k)]) < (1e-4 if k == 'weights' else 5e-5), (Ns, k)
    # other sample counts (2, 65, 100, 256) against the oracle; empty batch; N=1 is an error as in the reference
    rs = np.random.RandomState(0)
    for Ns in (2, 65, 100, 256):
        z = np.sort(rs.uniform(2, 6, (7, Ns)).astype(np.float32), -1)
        raw = rs.normal(size=(7, Ns, 4)).astype(np.float32) * 3
        rd = rs.normal(size=(7, 3)).astype(np.float32)
        ref = O.raw2outputs(raw, z, rd, None, True)
        out = raw2outputs(T(raw), T(z), T(rd), 0, True)
        for k, v, r in zip(('rgb', 'disp', 'acc', 'weights', 'depth'), out, ref):
            assert rel_err(N(v), r) < 1e-4, (Ns, k)
    out = raw2outputs(torch.empty((0, 64, 4), device=dev()), torch.empty((0, 64), device=dev()),
                      torch.empty((0, 3), device=dev()))
    assert out[0].shape == (0, 3)
    from nerfail_amd._lib import NerfailError
    with pytest.raises(NerfailError):
        raw2outputs(torch.zeros((3, 1, 4), device=dev()), torch.ones((3, 1), device=dev()), torch.ones((3, 3), device=dev()))


@pytest.mark.parametrize('Ns', [32, 96, 128, 160, 192, 224, 256])
def test_two_ray_composite_equals_one_ray_form_and_oracle(Ns):
    """Round 4: sample counts that are multiples of 32 run on composite2_kernel (two rays per wave, lane l of a half-wave owns
    samples l, l + 32, ...: coalesced, block-wise transmittance scan); every other count - and nerfail_composite_select(1) - on
    the one-ray-per-wave kernel. Same per-sample arithmetic, ray sums in another order: the two forms agree to rounding on
    every output incl. the argmax point (odd ray counts: the idle half-wave of the last wave; noise; both backgrounds; the
    point tensor given or formed from the ray), and both agree with the oracle."""
    from nerfail_amd.run_nerf import _composite
    from nerfail_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(Ns)
    for R in (1, 5, 64):
        z = np.sort(rs.uniform(2, 6, (R, Ns)).astype(np.float32), -1)
        raw = (rs.normal(size=(R, Ns, 4)) * 3).astype(np.float32)
        raw[:, :, 3] *= 4.0                                           # some saturated, some empty samples
        noise = rs.normal(size=(R, Ns)).astype(np.float32)
        rays = synth.ray_batch(R, seed=Ns + R)
        pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).astype(np.float32)
        for wb, nz, with_pts in ((True, None, False), (False, noise, True)):
            outs = {}
            for form in ('2', '1'):
                prev = lib.nerfail_composite_select(1 if form == '1' else 0)
                try:
                    outs[form] = [N(t) for t in _composite(T(raw), T(z), T(rays), T(nz) if nz is not None else None, wb,
                                                           T(pts) if with_pts else None, True)]
                finally:
                    lib.nerfail_composite_select(prev)
            ref = O.raw2outputs(raw, z, rays[:, 3:6], nz, wb)
            for k, a, b, r in zip(('rgb', 'disp', 'acc', 'weights', 'depth'), outs['2'], outs['1'], ref):
                assert rel_err(a, b) < 2e-6, (Ns, R, wb, k)                       # the two forms: summation order only
                # and the oracle. Per-sample weights: alpha = 1 - exp(-x) cancels for faint samples - one ulp of exp (numpy's
                # libm vs expf) is ~1e-4 of a 6e-4 weight; the composited maps average that out
                assert rel_err(a, r) < (3e-4 if k == 'weights' else 1e-4), (Ns, R, wb, k)
            w = outs['2'][3]
            best = w.argmax(1)                                                    # first maximum, as torch.argmax (NC:418)
            want = pts[np.arange(R), best]
            clear = np.sort(w, 1)[:, -1] - np.sort(w, 1)[:, -2] > 1e-6            # (a near-tie may resolve differently in the last bit)
            for form in ('2', '1'):
                got = outs[form][5]
                assert np.abs(got - want)[clear].max(initial=0.0) <= (0.0 if with_pts else 1e-6), (Ns, R, form)


def _check_samples(got, ref, u, bins):
    last_bin = (bins[:, -1] - bins[:, -2])[:, None] * 1.0001
    edge = u >= np.float32(0.999999)
    assert rel_err(np.where(edge, ref, got), ref) < 1e-5
    assert (np.abs(got - ref) <= last_bin)[edge].all()


def test_sample_pdf(golden):
    from nerfail_amd.run_nerf_helpers import sample_pdf
    g = golden('g5_sample_pdf')
    det = N(sample_pdf(T(g['bins']), T(g['weights']), 128, det=True))
    _check_samples(det, g['det'], np.broadcast_to(O.torch_linspace01(128), det.shape), g['bins'])
    rnd = N(sample_pdf(T(g['bins']), T(g['weights']), 128, det=False, u=T(g['u'])))
    _check_samples(rnd, g['rnd'], g['u'], g['bins'])
    # pytest=True draws from numpy seed 0 exactly like RH:215-223
    a = N(sample_pdf(T(g['bins']), T(g['weights']), 16, det=False, pytest=True))
    np.random.seed(0)
    uu = np.random.rand(48, 16).astype(np.float32)
    _check_samples(a, O.sample_pdf(g['bins'], g['weights'], 16, u=uu), uu, g['bins'])


def test_render_rays_cfg1(golden):
    from nerfail_amd import run_nerf as RN
    g = golden('g6_render_rays')
    sd, net = hip_nerf(4, 64, int(g['cfg1_seed']))
    r = RN.render_rays(T(g['cfg1_rays']), net, None, 64, retraw=True, white_bkgd=True)
    for k in ('rgb_map', 'disp_map', 'acc_map', 'raw'):
        assert rel_err(N(r[k]), g['cfg1_' + k]) < 1e-4, k
    assert 'rgb0' not in r and 'pts_max' not in r


def test_render_rays_cfg2(golden):
    from nerfail_amd import nerf_to_coord as NC, run_nerf as RN
    g = golden('g6_render_rays')
    _, coarse = hip_nerf(8, 256, int(g['cfg2_seed_coarse']))
    _, fine = hip_nerf(8, 256, int(g['cfg2_seed_fine']))
    keys = ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'pts_max')
    r = NC.render_rays(T(g['cfg2_rays']), coarse, None, 64, retraw=True, N_importance=128, network_fine=fine,
                       white_bkgd=True)
    for k in keys:
        assert rel_err(N(r[k]), g['cfg2_det_' + k]) < 1e-4, k
    # The LITERAL reading of "1e-4 rel" (VERDICT r4): max |a-b| / |b| over every non-zero reference entry, no floor - on EVERY
    # composited output, both passes, deterministic and perturbed (round 4 asserted it at 1e-3 on three keys only).
    def literal(rr, tag):
        for k in keys:
            e = plain_rel_err(N(rr[k]), g['cfg2_%s_%s' % (tag, k)])
            print('cfg2 %-4s %-8s rel_err %.2e  literal max |a-b|/|b| %.2e' % (tag, k, rel_err(N(rr[k]), g['cfg2_%s_%s' % (tag, k)]), e))
            assert e < 1e-4, (tag, k, e)
    literal(r, 'det')
    assert rel_err(N(r['raw']), g['cfg2_det_raw']) < 1e-2
    r = NC.render_rays(T(g['cfg2_rays']), coarse, None, 64, retraw=True, N_importance=128, network_fine=fine,
                       white_bkgd=True, perturb=1., t_rand=T(g['cfg2_t_rand']), u=T(g['cfg2_u']))
    for k in keys:
        assert rel_err(N(r[k]), g['cfg2_pert_' + k]) < 1e-4, k
    literal(r, 'pert')
    # the run_nerf flavour has no pts_max and the same maps
    r2 = RN.render_rays(T(g['cfg2_rays']), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True)
    assert 'pts_max' not in r2
    assert rel_err(N(r2['rgb_map']), g['cfg2_det_rgb_map']) < 1e-4


def test_render_wrapper(golden):
    from nerfail_amd import nerf_to_coord as NC, run_nerf as RN
    g = golden('g6_render_rays')
    _, coarse = hip_nerf(8, 256, 21)
    _, fine = hip_nerf(8, 256, 22)
    q = RN.FusedNetworkQuery(RN.get_embedder(10, 0)[0], RN.get_embedder(4, 0)[0])
    kw = dict(network_query_fn=q, perturb=0., N_importance=128, network_fine=fine, N_samples=64, network_fn=coarse,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
    rgb, disp, acc, pts_max, extras = NC.render(8, 8, g['render_K'], chunk=40, c2w=T(g['render_c2w']), near=2., far=6., **kw)
    assert rgb.shape == (8, 8, 3) and disp.shape == (8, 8) and pts_max.shape == (8, 8, 3)
    for v, gk in ((rgb, 'render_rgb'), (disp, 'render_disp'), (acc, 'render_acc'), (pts_max, 'render_pts_max'),
                  (extras['rgb0'], 'render_rgb0'), (extras['z_std'], 'render_z_std')):
        assert rel_err(N(v), g[gk]) < 1e-4, gk
    # rays= form (training path RN:776): same pixels given as explicit (rays_o, rays_d)
    ro, rd = RN.get_rays(8, 8, g['render_K'], T(g['render_c2w']))
    out = RN.render(8, 8, g['render_K'], chunk=1024, rays=(ro.reshape(-1, 3), rd.reshape(-1, 3)), near=2., far=6., **kw)
    assert rel_err(N(out[0]).reshape(8, 8, 3), g['render_rgb']) < 1e-4
    assert len(out) == 4 and set(out[3].keys()) == {'rgb0', 'disp0', 'acc0', 'z_std'}


def test_render_full_size_properties():
    """cfg2 sizes (64+128, D=8 W=256) on a 32 768-ray chunk of an 800x800 view: size-independent properties."""
    from nerfail_amd import nerf_to_coord as NC
    from nerfail_amd.run_nerf import ray_gen
    _, coarse = hip_nerf(8, 256, 21)
    _, fine = hip_nerf(8, 256, 22)
    focal, K = synth.lego_intrinsics(800, 800)
    c2w = synth.pose_spherical(-117., -30., 4.)[:3, :4]
    rays = ray_gen(800, 800, K, c2w, 2., 6., pix_begin=300 * 800, pix_count=32768)
    kw = dict(N_importance=128, network_fine=fine, white_bkgd=True, retraw=False)
    a = NC.render_rays(rays, coarse, None, 64, **kw)
    b = NC.render_rays(rays, coarse, None, 64, **kw)
    for k in a:                                            # deterministic: bitwise idempotent
        assert torch.equal(a[k].nan_to_num(7.), b[k].nan_to_num(7.)), k
    parts = [NC.render_rays(rays[s:s + 5000], coarse, None, 64, **kw) for s in range(0, 32768, 5000)]
    for k in a:                                            # chunk / shard invariance: bitwise
        cat = torch.cat([p[k] for p in parts], 0)
        assert torch.equal(a[k].nan_to_num(7.), cat.nan_to_num(7.)), k
    acc, rgb = N(a['acc_map']), N(a['rgb_map'])
    assert np.isfinite(rgb).all() and (acc >= 0).all() and (acc <= 1 + 1e-5).all()
    assert (rgb >= -1e-5).all() and (rgb <= 1 + 1e-5).all()          # white background composite stays in [0,1]
    assert (N(a['z_std']) >= 0).all()
    # spot-check 64 rays of the big chunk against the oracle
    sel = np.arange(0, 32768, 512)
    sc, sf = synth.nerf_state_dict(seed=21), synth.nerf_state_dict(seed=22)
    ref = O.render_rays(N(rays)[sel], sc, 64, 128, sf, white_bkgd=True)
    for k in ('rgb_map', 'acc_map', 'disp_map', 'pts_max', 'z_std'):
        assert rel_err(N(a[k])[sel], ref[k]) < 1e-4, k


def test_sample_fine_sorted_and_matches_oracle():
    from nerfail_amd import run_nerf as RN
    rays = synth.ray_batch(33, seed=9)
    sd, net = hip_nerf(4, 64, 5)
    rs = np.random.RandomState(1)
    u = rs.uniform(size=(33, 128)).astype(np.float32)
    t_rand = rs.uniform(size=(33, 64)).astype(np.float32)
    r = RN.render_rays(T(rays), net, None, 64, N_importance=128, white_bkgd=True, perturb=1., t_rand=T(t_rand), u=T(u),
                       retraw=True, want_pts_max=True)
    ref = O.render_rays(rays, sd, 64, 128, None, white_bkgd=True, t_rand=t_rand, u=u, D=4, W=64)
    for k in ('rgb_map', 'acc_map', 'z_std', 'rgb0', 'pts_max'):
        assert rel_err(N(r[k]), ref[k]) < 1e-4, k


@pytest.mark.parametrize('nc,nf', [(64, 128), (8, 5), (33, 70), (100, 300)])
@pytest.mark.parametrize('case', ['ordered_row_u', 'random_u', 'ties', 'descending_coarse'])
def test_sample_fine_merge_is_the_sort(case, nc, nf):
    """nerfail_sample_fine's merged row (RN:397 sort(cat(z_vals, z_samples))) is bit for bit numpy's sort of the same
    values, and pts = o + d z, through both of its branches: the binary-search merge (both halves ascending) and the rank
    sort (random u; a descending z_vals); ties between and inside the halves (zero weights: repeated samples)."""
    from nerfail_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(4)
    R = 37
    rays = synth.ray_batch(R, seed=3)
    zc = np.sort(rs.uniform(2, 6, size=(R, nc)).astype(np.float32), -1)
    w = rs.uniform(0, 1, size=(R, nc)).astype(np.float32)
    u, row = np.ascontiguousarray(O.torch_linspace01(nf), np.float32), 1
    if case == 'random_u':
        u, row = rs.uniform(size=(R, nf)).astype(np.float32), 0
    if case == 'ties':
        w[:, nc // 12:nc * 5 // 8] = 0.0        # flat cdf stretches: many equal samples
        w[3] = 0.0                              # a ray nothing was hit on
        zc[:, nc // 6:nc // 6 + 3] = zc[:, nc // 6:nc // 6 + 1]     # equal coarse depths
    if case == 'descending_coarse':
        zc = zc[:, ::-1].copy()
    zs, zf = torch.empty((R, nf), device=dev()), torch.empty((R, nc + nf), device=dev())
    pts, zstd = torch.empty((R, nc + nf, 3), device=dev()), torch.empty((R,), device=dev())
    rays_d, zc_d, w_d, u_d = T(rays), T(zc), T(w), T(u)          # named: the pointers must outlive the call
    _lib.check(lib.nerfail_sample_fine(_lib.dev(rays_d), R, _lib.dev(zc_d), _lib.dev(w_d), nc, _lib.dev(u_d), row, nf,
                                       _lib.dev(zs), _lib.dev(zf), _lib.dev(pts), _lib.dev(zstd), _lib.stream()))
    merged = np.sort(np.concatenate([zc, N(zs)], -1), -1)
    assert np.array_equal(N(zf), merged)
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    assert np.array_equal(N(pts), (d * merged[..., None]).astype(np.float32) + o)
    if case != 'descending_coarse':
        bins = 0.5 * (zc[:, 1:] + zc[:, :-1])
        ref = O.sample_pdf(bins, w[:, 1:-1], nf, u=None if row else u)
        # z_samples against the oracle: equal to rounding, except where a draw sits within an ulp of a cdf entry and lands
        # in the neighbouring bin (the cdf's last ulp differs between any two implementations of RH:205-207): rare, and by
        # less than a bin
        got, uu = N(zs), np.broadcast_to(u, ref.shape)
        off = np.abs(got - ref) > 1e-5 * np.abs(ref).max()
        assert off.mean() < 2e-3
        assert (np.abs(got - ref)[off] <= np.diff(bins, axis=-1).max() * 1.0001).all()
        if (nc, nf) == (64, 128):
            _check_samples(got, ref, uu, bins)                               # u = 1 may land one bin over (RH:239)


def test_cpu_tensors_are_moved_not_computed_on_cpu():
    """Inputs on the CPU are copied to the GPU; outputs always live on the GPU (there is no CPU path)."""
    from nerfail_amd.run_nerf import raw2outputs
    out = raw2outputs(torch.zeros(2, 64, 4), torch.linspace(2, 6, 64).repeat(2, 1), torch.ones(2, 3))
    assert out[0].is_cuda


def test_render_options_match_reference(golden):
    """Options the shipped configs leave at their defaults, against the reference's own output (fixture g14): lindisp,
    raw_noise_std > 0 with explicit draws, black background; the pytest=True numpy-seed-0 overrides; c2w_staticcam and
    a caller-provided ray batch through render()."""
    from nerfail_amd import nerf_to_coord as NC, run_nerf as RN
    g = golden('g14_render_options')
    _, coarse = hip_nerf(4, 64, int(g['seed_coarse']))
    _, fine = hip_nerf(4, 64, int(g['seed_fine']))
    keys = ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'pts_max')
    r = NC.render_rays(T(g['rays']), coarse, None, 64, retraw=True, lindisp=True, perturb=1., N_importance=128,
                       network_fine=fine, white_bkgd=False, raw_noise_std=1.0, t_rand=T(g['a_t_rand']), u=T(g['a_u']),
                       noise=T(g['a_noise0']), noise_fine=T(g['a_noise1']))
    for k in keys:
        assert rel_err(N(r[k]), g['a_' + k]) < 1e-4, k
    r = NC.render_rays(T(g['rays']), coarse, None, 64, retraw=True, perturb=1., N_importance=128, network_fine=fine,
                       white_bkgd=True, raw_noise_std=0.5, pytest=True)
    for k in keys:
        assert rel_err(N(r[k]), g['b_' + k]) < 1e-4, k
    q = RN.FusedNetworkQuery(RN.get_embedder(10, 0)[0], RN.get_embedder(4, 0)[0])
    kw = dict(network_query_fn=q, perturb=0., N_importance=128, network_fine=fine, N_samples=64, network_fn=coarse,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0., ndc=False, lindisp=False)
    rgb, disp, acc, pts_max, extras = NC.render(6, 6, g['c_K'], chunk=16, c2w=T(g['c_c2w']), c2w_staticcam=T(g['c_c2w_static']),
                                                near=2., far=6., **kw)
    for v, gk in ((rgb, 'c_rgb'), (disp, 'c_disp'), (acc, 'c_acc'), (pts_max, 'c_pts_max'), (extras['z_std'], 'c_z_std')):
        assert rel_err(N(v), g[gk]) < 1e-4, gk
    out = RN.render(6, 6, g['c_K'], chunk=3, rays=T(g['c_batch_rays']), near=2., far=6., **kw)
    assert tuple(out[0].shape) == (4, 3)
    for v, gk in ((out[0], 'c_rays_rgb'), (out[1], 'c_rays_disp'), (out[2], 'c_rays_acc'), (out[3]['rgb0'], 'c_rays_rgb0')):
        assert rel_err(N(v), g[gk]) < 1e-4, gk


def test_lds_streaming_kernel_equals_register_streamed_kernel():
    """The two exact-f32 forward kernels (mlp_lds.hip: weights through an LDS ring; mlp.hip: weights through registers)
    run the same FMA chains in the same order: their outputs must agree BITWISE, on a size that spans several rounds of
    the persistent grid, ragged tile, both network shapes, fused encoding and embedded input."""
    from nerfail_amd import _lib
    from nerfail_amd.run_nerf import _mlp_points
    lib = _lib.load()
    rs = np.random.RandomState(3)
    try:
        for (D, W, seed, R, Ns) in ((8, 256, 10, 2731, 64), (4, 64, 11, 517, 192)):
            _, net = hip_nerf(D, W, seed)
            pts = T(rs.uniform(-3, 3, size=(R, Ns, 3)).astype(np.float32))
            vd = rs.normal(size=(R, 3)).astype(np.float32)
            vd = T(vd / np.linalg.norm(vd, axis=1, keepdims=True))
            out = {}
            for which in (1, 2):
                lib.nerfail_mlp_fwd_select(which)
                out[which] = _mlp_points(net, pts, vd)
            assert torch.equal(out[1].view(torch.int32), out[2].view(torch.int32)), (D, W)
            assert float(out[1].abs().max()) > 0
    finally:
        lib.nerfail_mlp_fwd_select(0)


@pytest.mark.parametrize('D,skips', [(8, [4]), (8, [3]), (8, []), (6, [2]), (6, [3]), (4, [1]), (4, [2]), (2, [])])
def test_vgpr_form_kernel_all_depths_and_skip_positions(D, skips):
    """the W = 256 inference kernel keeps one activation array in arch VGPRs and issues its MFMAs by inline asm
    (mlp_lds.hip, lds_part<.., VG>); the (4, 64) case of the test above is NT = 2 and never compiles that form. Here every
    instantiation that does - nerf_mlp_fwd_lds_kernel<8, SKIP, false>, SKIP = 0 / 1 / 2 (no skip; skip into the first / the
    second layer of a pair) - at every even depth the ring covers: bitwise equal to the register-streamed kernel (all
    MFMAs compiler-issued) and within 1e-4 of the oracle. tools/check_mfma_hazards.py checks the same code object statically."""
    from nerfail_amd import _lib
    from nerfail_amd.run_nerf import _mlp_points
    from nerfail_amd.run_nerf_helpers import NeRF
    lib = _lib.load()
    rs = np.random.RandomState(100 + 10 * D + (skips[0] if skips else 9))
    sd = synth.nerf_state_dict(D=D, W=256, skips=tuple(skips), seed=200 + D)
    net = NeRF(D=D, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=list(skips), use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.requires_grad_(False).to(dev())
    R, Ns = 1031, 64                                     # 2 062 tiles: two rounds of the persistent grid + a ragged last round
    pts = rs.uniform(-3, 3, size=(R, Ns, 3)).astype(np.float32)
    vd = rs.normal(size=(R, 3)).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=1, keepdims=True)
    out = {}
    try:
        for which in (1, 2):
            lib.nerfail_mlp_fwd_select(which)
            out[which] = _mlp_points(net, T(pts), T(vd))
    finally:
        lib.nerfail_mlp_fwd_select(0)
    assert torch.equal(out[1].view(torch.int32), out[2].view(torch.int32))
    n = 64
    ref = O.run_network(sd, pts[:n], vd[:n], D=D, W=256, skips=tuple(s_ for s_ in skips if s_ < D - 1))
    assert rel_err(N(out[2][:n]), ref) < 1e-4


def test_lds_training_forward_saves_the_same_bits_as_the_register_kernel():
    """Round 3: the training forward runs on the LDS-ring kernel too (mlp_lds.hip, TRAIN). Everything it saves for the
    backward - encodings, every layer's post-ReLU tile, the feature tile, the views tile, the ReLU bit masks - and `raw`
    must equal the register-streamed nerf_mlp_fwd_kernel<NT, true> BITWISE (same FMA chains, same slots), incl. a
    ragged last tile and waves without a tile of their own."""
    from nerfail_amd import _lib, _train
    lib = _lib.load()
    rs = np.random.RandomState(4)
    try:
        for (D, W, seed, R, Ns) in ((8, 256, 12, 1500, 64), (8, 256, 13, 37, 5), (4, 64, 14, 301, 192)):
            _, net = hip_nerf(D, W, seed, requires_grad=True)
            pts = T(rs.uniform(-3, 3, size=(R, Ns, 3)).astype(np.float32))
            vd = rs.normal(size=(R, 3)).astype(np.float32)
            vd = T(vd / np.linalg.norm(vd, axis=1, keepdims=True))
            out = {}
            for which in (1, 2):
                lib.nerfail_mlp_fwd_select(which)
                n = _train.acts_floats(net, R * Ns)
                acts = torch.full((n,), float('nan'), device=dev())
                raw, acts = _train.mlp_fwd_train(net, pts, vd, acts=acts)
                out[which] = (raw.clone(), acts.clone())
            assert torch.equal(out[1][0].view(torch.int32), out[2][0].view(torch.int32)), (D, W, 'raw')
            a1, a2 = out[1][1].view(torch.int32), out[2][1].view(torch.int32)
            if (R * Ns) % 32 == 0:
                assert torch.equal(a1, a2), (D, W, 'acts', int((a1 != a2).sum()))
            else:   # the padding samples of the ragged last tile are whatever the clamped sample gives: compare whole tiles only
                per_tile = n // ((R * Ns + 31) // 32)
                full = (R * Ns) // 32 * per_tile
                assert torch.equal(a1[:full], a2[:full]), (D, W, 'acts of the full tiles')
    finally:
        lib.nerfail_mlp_fwd_select(0)


def test_points_formed_inside_the_kernel_equal_the_point_tensor_form():
    """the MLP kernel forms pts = o + d z from the packed ray and the depth (nerfail_mlp_fwd_rays), the
    sampling kernels stop writing [R,N,3], the composite forms pts_max from the ray. Every output must equal the point-tensor
    form BITWISE: raw (inference and training forward, both kernels), the saved activations, pts_max."""
    from nerfail_amd import _lib, _train
    from nerfail_amd import run_nerf as RN
    lib = _lib.load()
    _, net = hip_nerf(8, 256, 17, requires_grad=True)
    R, Ns = 333, 64
    rays = T(synth.ray_batch(R, seed=9))
    rs = np.random.RandomState(9)
    z = torch.empty((R, Ns), device=dev())
    pts = torch.empty((R, Ns, 3), device=dev())
    t_rand = T(rs.uniform(size=(R, Ns)).astype(np.float32))
    _lib.check(lib.nerfail_sample_coarse(_lib.dev(rays), R, _lib.dev(RN.linspace01(Ns, dev())), Ns, _lib.dev(t_rand), 0, _lib.dev(z),
                                         _lib.dev(pts), _lib.stream()))
    z2 = torch.empty_like(z)
    _lib.check(lib.nerfail_sample_coarse(_lib.dev(rays), R, _lib.dev(RN.linspace01(Ns, dev())), Ns, _lib.dev(t_rand), 0, _lib.dev(z2),
                                         None, _lib.stream()))                      # pts = NULL: z only
    assert torch.equal(z, z2)
    vd = rays[:, 8:11].contiguous()
    try:
        for which in (1, 2):
            lib.nerfail_mlp_fwd_select(which)
            with torch.no_grad():
                a = RN._mlp_points(net, pts, vd)
                b = RN._mlp_rays(net, rays, z)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), which
            n_acts = _train.acts_floats(net, R * Ns)         # (pre-filled: a few padding KB per tile are never written)
            ra, aa = _train.mlp_fwd_train(net, pts, vd, acts=torch.full((n_acts,), float('nan'), device=dev()))
            rb, ab = _train.mlp_fwd_train_rays(net, rays, z, acts=torch.full((n_acts,), float('nan'), device=dev()))
            assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(ra.view(torch.int32), a.view(torch.int32)), which
            full = (R * Ns) // 32 * (aa.numel() // ((R * Ns + 31) // 32))
            assert torch.equal(aa[:full].view(torch.int32), ab[:full].view(torch.int32)), which
    finally:
        lib.nerfail_mlp_fwd_select(0)
    out_p = RN._composite(a, z, rays, None, True, pts, True)
    out_r = RN._composite(a, z, rays, None, True, None, True)
    for u, v in zip(out_p, out_r):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))


def test_render_rays_on_a_pair_trained_by_the_reference(golden):
    """every other reference-held fixture uses seeded random-init networks (flat densities, no importance
    bin ever flips). g21 is a D=4 W=64 coarse + fine pair the REFERENCE trained for 2 000 of its own steps (RN:776-801) on the
    analytic sphere and then rendered itself on 4 096 rays, deterministic and perturbed, in fp32 and in fp64. The HIP path
    is held to the reference's fp32 outputs: the coarse pass on every ray, the whole path up to the number of rays the
    reference's own two precisions disagree on (conftest.check_against_trained_reference)."""
    from nerfail_amd import nerf_to_coord as NC
    from nerfail_amd.run_nerf_helpers import NeRF
    g = golden('g21_trained_pair')
    sc, sf, rays, t_rand, u = trained_pair_inputs(g)

    def net(sd):
        m = NeRF(D=4, W=64, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.requires_grad_(False).to(dev())
    coarse, fine = net(sc), net(sf)
    with torch.no_grad():
        r = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True)
        rp = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                            t_rand=T(t_rand), u=T(u))
    check_against_trained_reference(g, 'det', {k: N(v) for k, v in r.items()}, 'HIP')
    check_against_trained_reference(g, 'pert', {k: N(v) for k, v in rp.items()}, 'HIP')
    # the argmax point (NC:418-423) on the rays where nothing flipped: the same sample, hence the same point - except where two
    # neighbouring samples carry (nearly) the same weight and the argmax resolves the tie the other way: then the point moves by
    # one sample spacing along the ray (< 0.1 in world units here), and such rays must stay rare
    same = np.abs(N(r['rgb_map']) - g['det_rgb_map']).max(1) < 1e-5
    d = np.abs(N(r['pts_max']) - g['det_pts_max']).max(1)[same]
    print('argmax point: %d of %d unflipped rays differ by more than 1e-4 (worst %.3f)' % (int((d > 1e-4).sum()), d.size, d.max()))
    assert (d > 1e-4).mean() < 0.01 and d.max() < 0.1


def test_render_rays_headline_shape_on_trained_weights(golden):
    """the same judgement at the HEADLINE shape. g22 = a D=8 W=256 coarse + fine pair (RN:435-441, the
    shipped configs) trained for 2 000 steps on the analytic sphere (tests/golden/g22_weights.npz: input data), rendered by the
    REFERENCE on 4 096 rays, deterministic and perturbed, in fp32 and in fp64. The HIP path (the LDS-ring f32 MFMA kernel the
    headline number is measured on) is held to the reference's fp32 outputs: the coarse pass on every ray, the whole path up
    to the number of rays the reference's own two precisions disagree on (RH:200-243 flips, conftest)."""
    from nerfail_amd import nerf_to_coord as NC
    from nerfail_amd.run_nerf_helpers import NeRF
    g = golden('g22_trained_pair_d8')
    sc, sf, rays, t_rand, u = trained_pair_inputs(g)

    def net(sd):
        m = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.requires_grad_(False).to(dev())
    coarse, fine = net(sc), net(sf)
    with torch.no_grad():
        r = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True)
        rp = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True, perturb=1.,
                            t_rand=T(t_rand), u=T(u))
    check_against_trained_reference(g, 'det', {k: N(v) for k, v in r.items()}, 'HIP D8W256')
    check_against_trained_reference(g, 'pert', {k: N(v) for k, v in rp.items()}, 'HIP D8W256')
    same = np.abs(N(r['rgb_map']) - g['det_rgb_map']).max(1) < 1e-5
    d = np.abs(N(r['pts_max']) - g['det_pts_max']).max(1)[same]
    print('argmax point: %d of %d unflipped rays differ by more than 1e-4 (worst %.3f)' % (int((d > 1e-4).sum()), d.size, d.max()))
    assert (d > 1e-4).mean() < 0.01 and d.max() < 0.1
    # both forward kernels (LDS ring = default, register-streamed) give the same bits on trained weights too
    from nerfail_amd import _lib
    lib = _lib.load()
    try:
        lib.nerfail_mlp_fwd_select(1)
        with torch.no_grad():
            r1 = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True)
        lib.nerfail_mlp_fwd_select(2)
        with torch.no_grad():
            r2 = NC.render_rays(T(rays), coarse, None, 64, N_importance=128, network_fine=fine, white_bkgd=True)
    finally:
        lib.nerfail_mlp_fwd_select(0)
    for k in ('rgb_map', 'acc_map', 'rgb0', 'pts_max'):
        assert torch.equal(r1[k].view(torch.int32), r2[k].view(torch.int32)), k
