"""-m gpu: the dispatcher-registered ops torch.ops.nerfail_mi.* (nerfail_amd/ops.py): opcheck (schema,
fake tensor, autograd registration, AOT dispatch) on every op, their values against the golden vectors through the op
handles, autograd through the registered backward formulas, and a torch.compile trace."""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_err
from hiputil import T, N, dev, hip_nerf

pytestmark = pytest.mark.gpu


def _samples(golden):
    import nerfail_amd.ops  # noqa: F401
    g4, g5, g10, g11 = golden('g4_raw2outputs'), golden('g5_sample_pdf'), golden('g10_gauss_net'), golden('g11_igsm_step')
    rays = torch.zeros((32, 11), device=dev())
    rays[:, 3:6] = T(g4['N64_rays_d'])
    _, net = hip_nerf(4, 64, 11)
    rs = np.random.RandomState(0)
    pts = T(rs.uniform(-2, 2, (6, 16, 3)).astype(np.float32))
    vd = T(np.tile(np.array([[0.6, 0., 0.8]], np.float32), (6, 1)))
    focal, K = synth.lego_intrinsics(8, 8)
    c2w = synth.pose_spherical(30., -30., 4.)[:3, :4]
    S = T(synth.sphere_shell_points(300, seed=1))
    Q = T(synth.sphere_shell_points(50, seed=2))
    ones = lambda *s: torch.ones(s, device=dev())
    raw = T(g4['N64_raw'])
    return {
        'ray_gen': (torch.from_numpy(K), torch.from_numpy(c2w), rays, 8, 8, 2., 6., 5, 40),
        'composite': (raw.clone().requires_grad_(True), T(g4['N64_z']), rays, True),
        'composite_bwd': (raw, T(g4['N64_z']), rays, True, ones(32, 3), ones(32), ones(32), ones(32, 64), ones(32)),
        'sample_pdf': (T(g5['bins']), T(g5['weights']), T(g5['u'])),
        'mlp_fwd': (net.packed(), pts, vd, 4, 64, -1),
        'mlp_fwd_train': (net.packed(), pts, vd, 4, 64, -1),
        'mlp_bwd': _mlp_bwd_args(net, pts, vd),
        'sample_fine': (T(synth.ray_batch(32, seed=4)), T(g4['N64_z']), T(np.abs(g4['N64_raw'][..., 0]).astype(np.float32)),
                        torch.linspace(0., 1., 24, device=dev())),
        'knn8': (Q, S),
        'gauss_weight': (T(np.stack([np.abs(rs.normal(scale=0.02, size=(1, 4, 4, 8))), rs.randint(0, 48, (1, 4, 4, 8))], 1).astype(np.float32)), 0.02),
        'gauss_gather': (T(g10['s']).requires_grad_(True), T(g10['wi']), T(g10['ori']), 32.0),
        'gauss_gather_bwd': (T(g10['wi']), T(g10['ori']), T(g10['eps32_x']), T(g10['Gx']), T(g10['Gr']), 3 * 32 * 32, 32.0),
        'igsm_step': (T(g11['s']), T(g11['grad']), T(g11['s_init']), 2.0, 32.0, False),
    }


def _mlp_bwd_args(net, pts, vd):
    import nerfail_amd.ops as O
    from nerfail_amd import _train
    raw, acts = O.mlp_fwd_train(net.packed(), pts, vd, net.D, net.W, net._skip())
    d_raw = torch.randn_like(raw) * 1e-2
    return (net.packed(), _train.packed_T(net), acts, d_raw, net.D, net.W, net._skip())


def test_every_op_is_registered_and_passes_opcheck(golden):
    import nerfail_amd.ops as O
    samples = _samples(golden)
    assert set(samples) == set(O.ALL)
    for name in O.ALL:
        op = getattr(torch.ops.nerfail_mi, name).default
        tests = ('test_schema', 'test_faketensor', 'test_autograd_registration', 'test_aot_dispatch_dynamic')
        if name in ('gauss_gather_bwd', 'mlp_bwd', 'mlp_fwd_train'):   # float atomics (last bits differ run to run) / padding slots of
                                                                     # `acts` that no kernel writes: the AOT test compares outputs bit for bit
            tests = ('test_schema', 'test_faketensor', 'test_autograd_registration')
        args = samples[name]
        if name == 'composite':                     # rows 0..3 have acc = 0 -> disp = NaN by the reference's semantics (RN:299);
            args = (args[0].detach()[4:].clone().requires_grad_(True), args[1][4:], args[2][4:], args[3])   # opcheck compares without equal_nan
        res = torch.library.opcheck(op, args, test_utils=tests)
        assert all(v == 'SUCCESS' for v in res.values()), (name, res)


def test_op_values_and_registered_autograd(golden):
    import nerfail_amd.ops  # noqa: F401
    ops = torch.ops.nerfail_mi
    s = _samples(golden)
    g4, g5, g10, g11 = golden('g4_raw2outputs'), golden('g5_sample_pdf'), golden('g10_gauss_net'), golden('g11_igsm_step')
    rgb, disp, acc, w, depth = ops.composite(*s['composite'])
    assert rel_err(N(rgb), g4['N64_wb1_rgb']) < 1e-4 and rel_err(N(w), g4['N64_wb1_weights']) < 1e-4
    (rgb.sum() + acc.sum()).backward()                                   # registered backward formula -> composite_bwd
    raw = s['composite'][0]
    assert raw.grad is not None and torch.isfinite(raw.grad).all() and float(raw.grad.abs().max()) > 0
    from test_hip_nerf import _check_samples              # (u within an ulp of a bin edge may land one bin over, RH:239)
    _check_samples(N(ops.sample_pdf(*s['sample_pdf'])), g5['rnd'], g5['u'], g5['bins'])
    x, xr = ops.gauss_gather(*s['gauss_gather'])
    assert rel_err(N(x), g10['eps32_x']) < 1e-5 and rel_err(N(xr), g10['eps32_x_rgba']) < 1e-5
    ((x * T(g10['Gx'])).sum() + (xr * T(g10['Gr'])).sum()).backward()    # registered backward formula -> gauss_gather_bwd
    assert rel_err(N(s['gauss_gather'][0].grad), g10['eps32_grad_s']) < 1e-4
    assert np.array_equal(N(ops.igsm_step(*s['igsm_step'])), g11['out_targeted0'])
    d, i = ops.knn8(*s['knn8'])
    from oracle import knn as OK
    od, oi = OK.knn8(N(s['knn8'][0]), N(s['knn8'][1]))
    assert np.array_equal(N(d), od) and np.array_equal(N(i), oi.astype(np.float32))


def test_ops_trace_under_torch_compile(golden):
    """The ops are visible to the dispatcher: a function made of them compiles (fake tensors, no graph break on the ops)."""
    import nerfail_amd.ops  # noqa: F401
    s = _samples(golden)

    def f(spatial, wi, ori, s_init):
        x, xr = torch.ops.nerfail_mi.gauss_gather(spatial, wi, ori, 32.0)
        return torch.ops.nerfail_mi.igsm_step(spatial, xr.sum() * torch.ones_like(spatial), s_init, 2.0, 32.0, False)
    g10 = golden('g10_gauss_net')
    args = (T(g10['s']), T(g10['wi']), T(g10['ori']), T(g10['s']))
    eager = f(*args)
    comp = torch.compile(f, backend='aot_eager', fullgraph=True)(*args)
    assert torch.equal(eager, comp)


def test_training_ops_match_the_autograd_function_path():
    """torch.ops.nerfail_mi.mlp_fwd_train / mlp_bwd / sample_fine (the kernel-level halves of SURVEY 8b's nerf_mlp_fwd/bwd and
    merge_sorted) against the path render_rays takes in training (_train.mlp_fwd_train / mlp_backward, run_nerf's fine
    sampling): same raw bit for bit, parameter gradients equal up to the atomics' summation order."""
    import nerfail_amd.ops as O
    from nerfail_amd import _train
    _, net = hip_nerf(8, 256, 5)
    for p in net.parameters():
        p.requires_grad_(True)
    rs = np.random.RandomState(2)
    pts = T(rs.uniform(-2, 2, (40, 24, 3)).astype(np.float32))
    vd = T(rs.normal(size=(40, 3)).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)
    raw_a, acts_a = _train.mlp_fwd_train(net, pts, vd)
    raw_b, acts_b = O.mlp_fwd_train(net.packed(), pts, vd, net.D, net.W, net._skip())
    assert torch.equal(raw_a, raw_b)                    # (`acts` has padding slots no kernel writes: compared through the gradients)
    d_raw = T((rs.normal(size=(40, 24, 4)) * 1e-2).astype(np.float32))
    ga = _train._zero_grads(net)
    _train.mlp_backward(net, d_raw, acts_a, ga)
    gb = O.mlp_bwd(net.packed(), _train.packed_T(net), acts_b, d_raw, net.D, net.W, net._skip())
    assert len(ga) == len(gb) == 2 * net.D + 8
    for a, b, p in zip(ga, gb, _train.ordered_params(net)):
        assert tuple(b.shape) == tuple(p.shape)
        assert rel_err(N(b), N(a)) < 1e-4
    # sample_fine: the sorted row really is the sort, z_std the (biased) std of the new samples
    rays = T(synth.ray_batch(40, seed=6))
    zc = torch.sort(torch.rand((40, 64), device=dev()) * 4 + 2, -1)[0]
    w = torch.rand((40, 64), device=dev())
    zs, zf, pts_f, zstd = O.sample_fine(rays, zc, w, torch.linspace(0., 1., 128, device=dev()))
    assert torch.equal(zf, torch.sort(torch.cat([zc, zs], -1), -1)[0])
    assert rel_err(N(zstd), N(zs.std(-1, unbiased=False))) < 1e-4
    assert torch.equal(pts_f, rays[:, None, 3:6] * zf[..., None] + rays[:, None, 0:3])
