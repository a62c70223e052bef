"""-m gpu: deepfool (deepfool.py:10-111, the NeRFail inner loop) over the HIP gauss path vs the reference's own run
(fixture g12: same inputs, same stand-in classifier weights)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from hiputil import T, N, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag,target,over,iters', [('untargeted', None, 0.02, 6), ('targeted', 2, 0.02, 6),
                                                   ('untargeted_break', None, 1.0, 12), ('targeted_break', 2, 1.0, 12)])
def test_deepfool_matches_reference(golden, tag, target, over, iters):
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.deepfool import deepfool
    g = golden('g12_deepfool')
    w = T(g['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t()
    net = gauss_net(dev(), 0.02, Cls(), 'my_model', epsilon=None)
    # host tensors for the targeted case: deepfool uploads them once (the other cases pass device tensors)
    inp = (torch.from_numpy(g['s']), torch.from_numpy(g['wi']), torch.from_numpy(g['ori'])) if tag == 'targeted' \
        else (T(g['s']), T(g['wi']), T(g['ori']))
    rot, loop_i, ori_idx, cla_idx, s_new = deepfool(inp, 1.0, net, num_classes=8,
                                                    max_iter=iters, target_label=target, overshoot=over, m1=0.05, m2=0.5)
    assert loop_i == int(g[tag + '_loop_i'])
    assert int(ori_idx) == int(g[tag + '_ori_idx']) and int(cla_idx) == int(g[tag + '_cla_idx'])
    assert rel_err(N(rot), g[tag + '_rot']) < 1e-3            # 6-12 chained gradient steps in fp32
    assert rel_err(N(s_new), g[tag + '_s_new']) < 1e-3
    assert np.array_equal(N(s_new)[..., 3], g['s'][..., 3])   # alpha channel untouched


def _small_problem(seed=5, P=3, H=24, W=20, eps=None):
    import synth
    from nerfail_amd.GaussNet import create_gauss_w
    rs = np.random.RandomState(seed)
    s = rs.uniform(-40, 40, size=(P, H, W, 4)).astype(np.float32)
    s[..., 3] = rs.choice([0.0, 128.0, 255.0], size=(P, H, W))
    ori = synth.disc_alpha_image(1, H, W, seed=seed + 1)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(1, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, size=(1, H, W, 8)).astype(np.float32)
    wi, _ = create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    return T(s), wi, T(ori)


@pytest.mark.parametrize('eps', [None, 12.0])
def test_multi_rhs_backward_is_bitwise_the_single_backward(eps):
    """nerfail_gauss_bwd_csr_multi (all class gradients in one pass over the inverted index) vs one
    nerfail_gauss_bwd_csr call per right-hand side: identical bits, for 1..8 right-hand sides."""
    from nerfail_amd import _lib
    from nerfail_amd.GaussNet import gauss_gather, csr_for
    s, wi, ori = _small_problem(eps=eps)
    lib = _lib.load()
    st = s.clone().requires_grad_(True)
    x, x_rgba = gauss_gather(st, wi, ori, eps, None, True)
    n, B, P = s.numel() // 4, 1, ori.shape[1] * ori.shape[2]
    csr = csr_for(wi, n)
    rng = np.random.default_rng(0)
    for C in (1, 3, 8):
        J = T(rng.normal(size=(C, B * P, 4)).astype(np.float32))
        out = torch.empty((C, n, 4), device=dev())
        scratch = torch.empty((lib.nerfail_gauss_bwd_scratch_floats(B, P, C),), device=dev())
        _lib.check(lib.nerfail_gauss_bwd_csr_multi(_lib.dev(ori), _lib.dev(x.detach()), _lib.dev(J), C, _lib.dev(csr.row_ptr),
                                                   _lib.dev(csr.contrib), _lib.dev(csr.w_sorted), _lib.dev(csr.row_of), n, B, P,
                                                   -1.0 if eps is None else eps, _lib.dev(scratch), _lib.dev(out), _lib.stream()))
        for c in range(C):
            ref = torch.autograd.grad(x_rgba, st, grad_outputs=J[c].reshape(x_rgba.shape), retain_graph=True)[0]
            assert torch.equal(out[c].reshape(ref.shape), ref), (C, c)
    assert lib.nerfail_gauss_bwd_csr_multi(_lib.dev(ori), _lib.dev(x.detach()), _lib.dev(J), 9, _lib.dev(csr.row_ptr),
                                           _lib.dev(csr.contrib), _lib.dev(csr.w_sorted), _lib.dev(csr.row_of), n, B, P, -1.0, _lib.dev(scratch),
                                           _lib.dev(out), _lib.stream()) != 0


def test_logit_gradients_match_autograd():
    """gauss_net.logit_gradients (batched classifier backward + multi-RHS K11) vs torch.autograd.grad through forward(),
    one class at a time, with a small CNN (conv / ReLU / max-pool: the op set of the reference's classifier)."""
    from nerfail_amd.GaussNet import gauss_net
    s, wi, ori = _small_problem(seed=9)
    torch.manual_seed(3)
    cnn = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.ReLU(), torch.nn.MaxPool2d(2), torch.nn.Flatten(),
                              torch.nn.Linear(8 * 11 * 9, 8)).to(dev())
    cnn.requires_grad_(False)
    net = gauss_net(dev(), 0.02, cnn, 'my_model', epsilon=None)
    st = s.clone().requires_grad_(True)
    x, x_rgba, cla, _, _ = net(st, wi, ori)
    classes = [5, 0, 1, 2, 3, 4, 6, 7]
    G = net.logit_gradients(st, wi, x, x_rgba, cla, classes)
    for i, k in enumerate(classes):
        ref = torch.autograd.grad(cla[0, k], st, retain_graph=True)[0]
        assert rel_err(N(G[i]), N(ref)) < 1e-6, k


def test_deepfool_step_kernels():
    """K14 (nerfail_deepfool_norms / nerfail_deepfool_apply) against numpy: squared norms of the gradient differences
    (float64 reference, fixed-tree reduction => repeatable bits), the fused rot / clamp / alpha-restore update exactly."""
    from nerfail_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(14)
    for C, n in ((8, 3 * 37 * 41), (2, 5), (5, 70000)):
        G = (rng.normal(size=(C, n, 4)) * rng.uniform(0.1, 30, size=(C, 1, 1))).astype(np.float32)
        Gd = T(G)
        nb = lib.nerfail_deepfool_norms_scratch_bytes(C, n)
        assert nb > 0
        scratch = torch.empty((nb,), dtype=torch.uint8, device=dev())
        out = []
        for _ in range(2):
            norms2 = torch.empty((C - 1,), device=dev())
            _lib.check(lib.nerfail_deepfool_norms(_lib.dev(Gd), C, n, _lib.dev(scratch), nb, _lib.dev(norms2), _lib.stream()))
            out.append(N(norms2))
        assert np.array_equal(out[0], out[1])
        ref = ((G[1:].astype(np.float64) - G[0:1].astype(np.float64)) ** 2).reshape(C - 1, -1).sum(1)
        assert np.all(np.abs(out[0] - ref) <= 2e-6 * ref)
        s0 = rng.uniform(-300, 300, size=(n, 4)).astype(np.float32)
        rot = rng.normal(size=(n, 4)).astype(np.float32)
        k, sc, over = C - 1, np.float32(0.37), 0.02
        rot_d, new_s = T(rot.copy()), torch.empty((n, 4), device=dev())
        best_d, sc_d, s0_d = torch.tensor([k], dtype=torch.int32, device=dev()), torch.tensor([sc], device=dev()), T(s0)
        _lib.check(lib.nerfail_deepfool_apply(_lib.dev(Gd), C, n, _lib.dev(best_d), _lib.dev(sc_d), over, _lib.dev(s0_d),
                                              _lib.dev(rot_d), _lib.dev(new_s), _lib.stream()))
        rot_ref = rot + sc * (G[k] - G[0])
        s_ref = np.clip(s0 + np.float32(over) * rot_ref, -255, 255).astype(np.float32)
        s_ref[:, 3] = s0[:, 3]
        assert np.array_equal(N(rot_d), rot_ref) and np.array_equal(N(new_s), s_ref)
        # scale 0 = nothing chosen: rot untouched even if the gradients hold inf
        Gbad = Gd.clone()
        Gbad[k, 0, 0] = float('inf')
        rot_d, zero_d = T(rot.copy()), torch.zeros(1, device=dev())
        _lib.check(lib.nerfail_deepfool_apply(_lib.dev(Gbad), C, n, _lib.dev(best_d), _lib.dev(zero_d), over, _lib.dev(s0_d),
                                              _lib.dev(rot_d), _lib.dev(new_s), _lib.stream()))
        assert np.array_equal(N(rot_d), rot)
    assert lib.nerfail_deepfool_norms_scratch_bytes(1, 10) == 0 and lib.nerfail_deepfool_norms_scratch_bytes(9, 10) == 0
    assert lib.nerfail_deepfool_norms(_lib.dev(Gd), 5, 70000, _lib.dev(scratch), 8, _lib.dev(norms2), _lib.stream()) != 0


def test_deepfool_class_counts_outside_the_multi_rhs_range():
    """the multi-RHS pass takes 2..8 logits. A classifier whose prediction lies outside range(num_classes)
    leaves num_classes competitors (9 logits at num_classes = 8), num_classes = 1 leaves one: both must take the
    class-by-class branch like the reference instead of raising."""
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.deepfool import deepfool
    s, wi, ori = _small_problem(seed=9)
    rs = np.random.RandomState(3)
    w = T((rs.normal(size=(10, 48)) * 0.05).astype(np.float32))
    bias = torch.zeros(10, device=dev())
    bias[9] = 1e4                                               # prediction = class 9 >= num_classes = 8 (logits are ~1e2)

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t() + bias
    net = gauss_net(dev(), 0.02, Cls(), 'my_model', epsilon=None)
    rot, loop_i, ori_idx, cla_idx, s_new = deepfool((s, wi, ori), 1.0, net, num_classes=8, max_iter=2, m1=0.05, m2=0.5)
    assert int(ori_idx) == 9 and loop_i == 2 and torch.isfinite(s_new).all()
    rot, loop_i, _, _, s_new = deepfool((s, wi, ori), 1.0, net, num_classes=1, max_iter=1, m1=0.05, m2=0.5)
    assert loop_i == 1 and torch.isfinite(s_new).all()
