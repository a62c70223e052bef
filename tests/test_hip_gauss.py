"""-m gpu: gauss path (K9-K12) through the C ABI against the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

import synth
from conftest import rel_err
from hiputil import T, N, dev
from oracle import gauss as OG

pytestmark = pytest.mark.gpu


def test_create_gauss_w(golden):
    from nerfail_amd.GaussNet import create_gauss_w
    g = golden('g9_gauss_w')
    i_w, dist = create_gauss_w(dev(), 0.02)(T(g['dist_and_index']))
    assert rel_err(N(i_w)[:, 0], g['i_w'][:, 0]) < 1e-5
    assert np.array_equal(N(i_w)[:, 1], g['i_w'][:, 1])
    assert np.array_equal(N(dist), g['dist'])
    assert (N(i_w)[0, 0, 0, 0] == 0).all()


def test_gauss_get_r_and_gauss_get_img(golden):
    """GN:189-337: `from model.GaussNet import gauss_net, gauss_get_r, gauss_get_img` (attack_NeRFail.py:23)
    works against the mirror; outputs vs fixture g20 (the reference run) and vs the oracle, gradients vs torch autograd of the
    same formulas."""
    from nerfail_amd.GaussNet import gauss_net, gauss_get_r, gauss_get_img      # noqa: F401  (the reference's import line)
    g = golden('g20_gauss_get')
    w = T(golden('g10_gauss_net')['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t()
    get_r = gauss_get_r(dev(), 0.02, Cls(), 'my_model')
    s = T(g['s']).requires_grad_(True)
    r = get_r(s, T(g['dist_and_index']))
    assert rel_err(N(r), g['r']) < 1e-5
    assert rel_err(N(r), OG.gauss_get_r(g['s'], g['dist_and_index'], 0.02)[0]) < 1e-5
    assert (N(r)[0, 0, 0] == 0).all()
    assert abs(get_r.epsilon_3d_max - float(g['eps3d_max'])) < 1e-3 * abs(float(g['eps3d_max']))
    assert abs(get_r.epsilon_3d_min - float(g['eps3d_min'])) < 1e-3 * abs(float(g['eps3d_min']))
    get_r.epsilon_3d_zero()
    assert get_r.epsilon_3d_max == 0 and get_r.epsilon_3d_min == 0
    # d r / d s: r is linear in s with the K9 weights as coefficients
    Gx = T(np.random.RandomState(3).normal(size=g['r'].shape).astype(np.float32))
    (r * Gx).sum().backward()
    wi = OG.create_gauss_w(g['dist_and_index'], 0.02)[0]
    ref = OG.gauss_backward(g['s'], wi, np.zeros_like(g['ori']), N(Gx), np.zeros_like(N(Gx)), None)
    assert rel_err(N(s.grad), ref) < 1e-4

    get_img = gauss_get_img(dev(), 0.02, Cls(), 'my_model')
    rt = T(g['r']).requires_grad_(True)
    r_out, x_rgba, cla, ori, ori_cla = get_img(T(g['ori']), rt)
    assert r_out is rt
    assert rel_err(N(x_rgba), g['x_rgba']) < 1e-6
    assert np.array_equal(N(x_rgba), OG.gauss_get_img(g['ori'], g['r']))         # same three roundings
    assert tuple(cla.shape) == (2, 8) and tuple(ori_cla.shape) == (2, 8) and torch.equal(ori, T(g['ori']))
    # gradient of the composite w.r.t. r against torch autograd of GN:309-319
    Gr = T(np.random.RandomState(4).normal(size=g['r'].shape).astype(np.float32))
    (x_rgba * Gr).sum().backward()
    r2 = T(g['r']).requires_grad_(True)
    o = T(g['ori'])
    x2 = torch.where(o[..., 3:4] > 0, o[..., :3] + r2[..., :3] * (r2[..., 3:4] / 255), torch.zeros_like(o[..., :3]))
    (torch.cat([x2, o[..., 3:4]], -1) * Gr).sum().backward()
    assert rel_err(N(rt.grad), N(r2.grad)) < 1e-5


@pytest.mark.parametrize('det', [True, False])
@pytest.mark.parametrize('tag,eps', [('epsNone_', None), ('eps32_', 32.0)])
def test_gauss_forward_backward(golden, tag, eps, det):
    from nerfail_amd.GaussNet import gauss_gather
    g = golden('g10_gauss_net')
    s = T(g['s']).requires_grad_(True)
    mm = torch.zeros(2, device=dev())
    x, x_rgba = gauss_gather(s, T(g['wi']), T(g['ori']), eps, mm, deterministic=det)
    assert rel_err(N(x), g[tag + 'x']) < 1e-5
    assert rel_err(N(x_rgba), g[tag + 'x_rgba']) < 1e-5
    assert abs(float(mm[1]) - float(g[tag + 'eps3d_max'])) < 1e-3 * abs(float(mm[1]))
    assert abs(float(mm[0]) - float(g[tag + 'eps3d_min'])) < 1e-3 * abs(float(mm[0]))
    ((x * T(g['Gx'])).sum() + (x_rgba * T(g['Gr'])).sum()).backward()
    assert rel_err(N(s.grad), g[tag + 'grad_s']) < 1e-4     # float atomics: order-dependent last bits


@pytest.mark.parametrize('tag,eps', [('epsNone_', None), ('eps32_', 32.0)])
def test_gauss_net_module_end_to_end(golden, tag, eps):
    """gauss_net.forward with the fixture's stand-in classifier: logits and d CE / d s vs the reference's autograd."""
    from nerfail_amd.GaussNet import gauss_net
    g = golden('g10_gauss_net')
    w = T(g['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t()
    net = gauss_net(dev(), 0.02, Cls(), 'my_model', epsilon=eps)
    s = T(g['s']).requires_grad_(True)
    x, x_rgba, cla, ori, ori_cla = net(s, T(g['wi']), T(g['ori']))
    assert rel_err(N(cla), g[tag + 'cla']) < 1e-4
    torch.nn.functional.cross_entropy(cla, torch.full((2,), 4, dtype=torch.long, device=dev())).backward()
    assert rel_err(N(s.grad), g[tag + 'ce_grad_s']) < 1e-4
    assert net.epsilon_3d_max > 0 and net.epsilon_3d_min < 0
    net.epsilon_3d_zero()
    assert net.epsilon_3d_max == 0


def test_igsm_step(golden):
    from nerfail_amd.attack import igsm_step
    g = golden('g11_igsm_step')
    for targeted in (False, True):
        out = igsm_step(T(g['s']), T(g['grad']), T(g['s_init']), 2.0, 32.0, targeted)
        assert np.array_equal(N(out), g['out_targeted%d' % int(targeted)])


def test_gauss_full_size_properties():
    """cfg3 sizes: P=3 base views, 800x800 pixels, B=2 views. Linearity of x in s, adjointness of the backward."""
    from nerfail_amd.GaussNet import gauss_gather, create_gauss_w
    rs = np.random.RandomState(0)
    P, B, H, W = 3, 2, 800, 800
    idx = rs.randint(0, P * H * W, size=(B, H, W, 8)).astype(np.float32)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    wi, _ = create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    ori = T(synth.disc_alpha_image(B, H, W, seed=3))
    s1 = T(rs.uniform(-20, 20, (P, H, W, 4)).astype(np.float32))
    s2 = T(rs.uniform(-20, 20, (P, H, W, 4)).astype(np.float32))
    x1, _ = gauss_gather(s1, wi, ori, None)
    x2, _ = gauss_gather(s2, wi, ori, None)
    x12, _ = gauss_gather(s1 + s2, wi, ori, None)
    assert np.abs(N(x12) - N(x1 + x2)).max() < 1e-4                        # x is linear in s (|x| ~ 20, fp32)
    # <G, J ds> == <J^T G, ds> for the linear part (x only)
    G = T(rs.normal(size=(B, H, W, 4)).astype(np.float32))
    s = s1.clone().requires_grad_(True)
    x, _ = gauss_gather(s, wi, ori, None)
    (x * G).sum().backward()
    # deterministic (inverted-index) backward: bitwise reproducible, and equal to the atomic form to rounding
    s_b = s1.clone().requires_grad_(True)
    xb, _ = gauss_gather(s_b, wi, ori, None)
    (xb * G).sum().backward()
    assert torch.equal(s.grad, s_b.grad)
    s_c = s1.clone().requires_grad_(True)
    xc, _ = gauss_gather(s_c, wi, ori, None, deterministic=False)
    (xc * G).sum().backward()
    assert rel_err(N(s_c.grad), N(s.grad)) < 1e-4
    lhs = float((x2.double() * G.double()).sum())
    rhs = float((s.grad.double() * s2.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), abs(rhs), 1.0) * 10
    # weights of every pixel sum to <= 1 and are non-negative
    wsum = N(wi[:, 0].sum(-1))
    assert (wsum <= 1 + 1e-5).all() and (N(wi[:, 0]) >= 0).all()
    # spot check 1 row of pixels against the oracle
    xs, xr, _ = OG.gauss_forward(N(s1), N(wi)[:, :, 400:401], N(ori)[:, 400:401], 32.0)
    xh, xrh = gauss_gather(s1, wi[:, :, 400:401].contiguous(), ori[:, 400:401].contiguous(), 32.0)
    assert rel_err(N(xh), xs) < 1e-5 and rel_err(N(xrh), xr) < 1e-5


def test_cached_original_logits_are_identical():
    """gauss_net.cache_ori_cla: same 5-tuple as the reference-shaped forward, the second call reuses the
    stored logits of the unperturbed images, a modified image tensor invalidates them."""
    from nerfail_amd.GaussNet import gauss_net, create_gauss_w
    rs = np.random.RandomState(3)
    P, B, H, W = 3, 2, 12, 12
    s = T(rs.uniform(-20, 20, (P, H, W, 4)).astype(np.float32))
    ori = T(synth.disc_alpha_image(B, H, W, seed=2))
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, (B, H, W, 8)).astype(np.float32)
    wi, _ = create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    torch.manual_seed(1)
    calls = []

    class Cls(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3 * 4 * 4, 8)

        def forward(self, x):
            calls.append(x.shape[0])
            return self.lin(torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1))
    net = gauss_net(dev(), 0.02, Cls().to(dev()), 'my_model', epsilon=None)
    ref = net(s, wi, ori)
    net.cache_ori_cla = True
    calls.clear()
    a = net(s, wi, ori)
    b = net(s, wi, ori)
    assert len(calls) == 3                                   # 2 + 1: the second forward skipped the original images
    for r, x, y in zip(ref, a, b):
        assert torch.equal(r, x) and torch.equal(r, y)
    ori.mul_(0.5)                                            # in-place change -> new version -> recomputed
    calls.clear()
    c = net(s, wi, ori)
    assert len(calls) == 2 and not torch.equal(c[4], ref[4])


def test_backward_with_very_long_rows():
    """A perturbation row that is the neighbour of thousands of pixels (kLongRow = 256 in gauss_csr.hip) takes the
    cooperative whole-wave path of the inverted-index backward: result vs the oracle, bitwise repeatable, and the
    multi-RHS kernel still bitwise equal to the single one."""
    from nerfail_amd import _lib
    from nerfail_amd.GaussNet import gauss_gather, create_gauss_w, csr_for
    rs = np.random.RandomState(21)
    P, B, H, W = 2, 2, 40, 36
    Ns = P * H * W
    s = rs.uniform(-30, 30, (P, H, W, 4)).astype(np.float32)
    s[..., 3] = 255.0
    ori = synth.disc_alpha_image(B, H, W, seed=5)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, Ns, (B, H, W, 8))
    hot = rs.uniform(size=idx.shape)
    idx[hot < 0.25] = 7                       # ~5 800 contributions to row 7
    idx[(hot >= 0.25) & (hot < 0.30)] = Ns - 1    # ~1 100 to the last row
    idx[(hot >= 0.30) & (hot < 0.32)] = 1000      # ~460 to a row in the middle
    wi_np, _ = OG.create_gauss_w(np.stack([dist, idx.astype(np.float32)], 1))
    G = rs.normal(size=(B, H, W, 4)).astype(np.float32)
    wi, oriT = T(wi_np), T(ori)
    grads = []
    for _ in range(2):
        st = T(s).requires_grad_(True)
        x, xr = gauss_gather(st, wi, oriT, 32.0, None, True)
        (xr * T(G)).sum().backward()
        grads.append(N(st.grad))
    assert np.array_equal(grads[0], grads[1])
    ref = OG.gauss_backward(s, wi_np, ori, np.zeros_like(G), G, 32.0)
    assert np.abs(grads[0] - ref).max() <= 2e-5 * np.abs(ref).max()
    # multi-RHS: slice c of one launch == a single launch with that right-hand side
    lib = _lib.load()
    n, Pp = Ns, H * W
    csr = csr_for(wi, n)
    J = T(rs.normal(size=(3, B * Pp, 4)).astype(np.float32))
    out = torch.empty((3, n, 4), device=dev())
    scratch = torch.empty((lib.nerfail_gauss_bwd_scratch_floats(B, Pp, 3),), device=dev())
    st = T(s).requires_grad_(True)
    x, xr = gauss_gather(st, wi, oriT, 32.0, None, True)
    xs = x.detach()
    _lib.check(lib.nerfail_gauss_bwd_csr_multi(_lib.dev(oriT), _lib.dev(xs), _lib.dev(J), 3, _lib.dev(csr.row_ptr),
                                               _lib.dev(csr.contrib), _lib.dev(csr.w_sorted), _lib.dev(csr.row_of), n, B, Pp, 32.0,
                                               _lib.dev(scratch), _lib.dev(out), _lib.stream()))
    sc1 = torch.empty((lib.nerfail_gauss_bwd_scratch_floats(B, Pp, 1),), device=dev())
    for c in range(3):                        # the SAME (batch) index, one right-hand side per call: identical bits
        single = torch.empty((n, 4), device=dev())
        _lib.check(lib.nerfail_gauss_bwd_csr(_lib.dev(oriT), _lib.dev(xs), None, _lib.dev(J[c]), _lib.dev(csr.row_ptr),
                                             _lib.dev(csr.contrib), _lib.dev(csr.w_sorted), _lib.dev(csr.row_of), n, B, Pp, 32.0,
                                             _lib.dev(sc1), 0, _lib.dev(single), _lib.stream()))
        assert torch.equal(out[c], single), c
        auto = torch.autograd.grad(xr, st, grad_outputs=J[c].reshape(xr.shape), retain_graph=True)[0]   # per-view indices: other order
        assert rel_err(N(auto).reshape(-1, 4), N(single)) < 1e-5, c


@pytest.mark.parametrize('n_total', [512 * 5 + 1, 512 * 5 + 7, 512 * 5 + 8, 512 * 5 + 9, 512 * 5 + 63, 512 * 5 + 64, 512 * 5 + 65,
                                     512 * 6 - 1, 512 * 6, 511, 8, 1])
def test_segmented_reduce_on_adversarial_row_layouts(n_total):
    """Round 4 rebuilt the inverted-index reduce (a lane owns 8 consecutive entries, one cross-lane scan per 512-entry chunk,
    an LDS-transposed route for full chunks and a direct one for the partial last chunk). The index of a view is the list of
    its non-zero-weight contributions sorted by row, so the row layout can be dictated exactly: a row of exactly 3 chunks, rows
    that end ON a chunk boundary, rows of exactly one lane (8 entries) aligned and misaligned, a 2 000-entry row across
    several chunks, 600 single-entry rows in sequence, and every kind of partial tail (n_total). Deterministic gradient vs the
    oracle's scatter-add, bitwise repeatable, equal to the float-atomic form to rounding."""
    from nerfail_amd.GaussNet import gauss_gather
    from nerfail_amd import GaussNet as G
    rs = np.random.RandomState(n_total)
    H, W, P = 24, 20, 3                                         # 480 pixels x 8 slots = 3 840 contribution slots
    Ns = P * H * W
    pattern = [1536, 1, 511, 8, 8, 8, 3, 8, 8, 5, 2000] + [1] * 600 + [9, 7, 512, 512, 17]
    lens, left = [], n_total
    for L_ in pattern:
        if left <= 0:
            break
        lens.append(min(L_, left))
        left -= lens[-1]
    assert sum(lens) == n_total <= H * W * 8
    rows = np.cumsum(rs.randint(1, 4, size=len(lens)))          # ascending row ids with empty rows in between
    assert rows[-1] < Ns
    slot_row = np.repeat(rows, lens)                            # sorted entry order == slot order (ascending contribution id)
    idx = np.zeros(H * W * 8, np.float32)
    w = np.zeros(H * W * 8, np.float32)
    idx[:n_total] = slot_row
    w[:n_total] = rs.uniform(0.1, 1.0, n_total)
    idx[n_total:] = rs.randint(0, Ns, H * W * 8 - n_total)      # weight 0: dropped from the index
    wi_np = np.stack([w.reshape(1, H, W, 8), idx.reshape(1, H, W, 8)], 1).astype(np.float32)
    s = rs.uniform(-30, 30, (P, H, W, 4)).astype(np.float32)
    s[..., 3] = 255.0
    ori = synth.disc_alpha_image(1, H, W, seed=9)
    ori[..., 3] = 255.0                                         # every pixel opaque: every gradient passes
    Gr = rs.normal(size=(1, H, W, 4)).astype(np.float32)
    G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()
    grads = {}
    for det in (True, True, False):
        st = T(s).requires_grad_(True)
        x, xr = gauss_gather(st, T(wi_np), T(ori), None, None, det)
        (xr * T(Gr)).sum().backward()
        grads.setdefault(det, []).append(N(st.grad))
    assert np.array_equal(grads[True][0], grads[True][1])       # fixed order: the same bits
    ref = OG.gauss_backward(s, wi_np, ori, np.zeros_like(Gr), Gr, None)
    scale = np.abs(ref).max()
    assert np.abs(grads[True][0] - ref).max() <= 2e-5 * scale
    assert np.abs(grads[False][0] - ref).max() <= 2e-5 * scale
    vi = G.view_indices(T(wi_np), Ns)[0]
    assert vi.n_entries == n_total and vi.n_rows == len(lens)
    G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()


def test_cfg3_loop_matches_reference_iterates(golden):
    """BASELINE configs[2] at fixture size (g15): 20 iterations x 2 batches of 8 views, sequential update, through
    attack.nerfail_s_loop (HIP gauss forward, deterministic inverted-index backward, sign-step kernel) against the
    iterates of the reference's gauss_net + the re-issued AS:352-392 step. A sign step can differ from the reference only
    where the gradient is at rounding level (the fixture's smallest nonzero |grad| is 3e-8 of a ~1e-3 typical value):
    the differing elements are counted and bounded, each by one step of a."""
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.attack import nerfail_s_loop
    g = golden('g15_cfg3_loop')
    P, H, W, NB, B, ITERS = [int(v) for v in g['shape']]
    w = T(g['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t()
    net = gauss_net(dev(), 0.02, Cls(), 'my_model', epsilon=None)
    wi, ori = T(g['wi']), T(g['ori'])
    batches = [(wi[b * B:(b + 1) * B].contiguous(), ori[b * B:(b + 1) * B].contiguous()) for b in range(NB)]
    ref = g['iterates_rgb_int8'].astype(np.float32)
    seen = []

    def on_iter(it, b, s, loss):
        seen.append((N(s), float(loss)))
    s0 = T(g['s0'])
    s_end = nerfail_s_loop(net, s0, s0, batches, torch.tensor(int(g['label']), device=dev()), ITERS,
                           float(g['a']), float(g['epsilon']), False, on_iter=on_iter)
    assert len(seen) == NB * ITERS
    worst = 0.0
    for step, (s, loss) in enumerate(seen):
        diff = s[..., :3] != ref[step]
        worst = max(worst, float(diff.mean()))
        assert np.abs(s[..., :3] - ref[step]).max() <= 2 * float(g['a']), step
        assert np.array_equal(s[..., 3], g['s0'][..., 3]), step                    # alpha untouched
        assert abs(loss - g['losses'][step]) <= 2e-4 * abs(g['losses'][step]), (step, loss, g['losses'][step])
    print('HIP vs reference cfg3 iterates: worst fraction of differing elements %.2e' % worst)
    assert worst < 2e-3
    assert np.abs(N(s_end)[..., :3]).max() <= float(g['epsilon'])


def test_gauss_gather_rejects_mismatched_shapes():
    """ori_img / spatial_rgb shapes are validated before any kernel indexes them as float4 arrays."""
    from nerfail_amd.GaussNet import gauss_gather
    wi = torch.zeros((2, 2, 6, 5, 8), device=dev())
    s = torch.zeros((3, 6, 5, 4), device=dev())
    with pytest.raises(ValueError, match='ori_img'):
        gauss_gather(s, wi, torch.zeros((2, 6, 4, 4), device=dev()))
    with pytest.raises(ValueError, match='ori_img'):
        gauss_gather(s, wi, torch.zeros((1, 6, 5, 4), device=dev()))
    with pytest.raises(ValueError, match='spatial_rgb'):
        gauss_gather(torch.zeros((3, 6, 5, 3), device=dev()), wi, torch.zeros((2, 6, 5, 4), device=dev()))
    x, xr = gauss_gather(s, wi, torch.zeros((2, 6, 5, 4), device=dev()))
    assert x.shape == (2, 6, 5, 4)


def test_per_view_index_cache_survives_fresh_tensors_and_shuffling(golden, tmp_path):
    """the attack loop draws batches from a DataLoader - every iteration a NEW tensor with the views in
    a new composition. The inverted indices are per VIEW, keyed by content fingerprint (or by caller-supplied ids): fresh
    copies and shuffled batches must not rebuild anything, the gradient must equal the atomic form's, and an index stored
    to disk must reproduce the bits."""
    from nerfail_amd import GaussNet as G
    g = golden('g10_gauss_net')
    wi, ori, s0 = T(g['wi']), T(g['ori']), T(g['s'])
    built = []
    orig_init = G.ViewIndex.__init__

    def counting_init(self, wi_view=None, Ns=None, state=None):
        if state is None:
            built.append(1)
        orig_init(self, wi_view, Ns, state)
    G.ViewIndex.__init__ = counting_init
    try:
        G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()

        def grad(wi_b, ori_b, Gr, view_ids=None, det=True):
            s = s0.clone().requires_grad_(True)
            x, xr = G.gauss_gather(s, wi_b, ori_b, 32.0, None, det, view_ids)
            (xr * Gr).sum().backward()
            return s.grad
        Gr = T(g['Gr'])
        ref = grad(wi, ori, Gr, det=False)
        a = grad(wi.clone(), ori, Gr)                           # fresh tensor #1: builds 2 view indices
        assert len(built) == 2
        b = grad(wi.clone(), ori, Gr)                           # fresh tensor #2: same content -> cache hits
        assert len(built) == 2 and torch.equal(a, b)
        assert rel_err(N(a), N(ref)) < 1e-4
        perm = [1, 0]                                           # shuffled composition: still no rebuild
        c = grad(wi[perm].contiguous(), ori[perm].contiguous(), Gr[perm].contiguous())
        assert len(built) == 2
        assert rel_err(N(c), N(ref)) < 1e-4                     # (view order changes the summation order, not the sum)
        d = grad(wi.clone(), ori, Gr, view_ids=[7, 11])         # caller-named views: their own keys
        assert len(built) == 4 and torch.equal(d, a)
        grad(wi.clone(), ori, Gr, view_ids=[7, 11])
        assert len(built) == 4
        # persisted per-view index: save, drop the cache, load, same bits
        vi = G.view_indices(wi, s0.numel() // 4, [7, 11])
        for k, v in zip((7, 11), vi):
            v.save(str(tmp_path / ('%d.idx.pth' % k)))
        G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()
        for k in (7, 11):
            G.register_view_index(k, G.ViewIndex.load(str(tmp_path / ('%d.idx.pth' % k))))
        e = grad(wi.clone(), ori, Gr, view_ids=[7, 11])
        assert len(built) == 4 and torch.equal(e, a)
    finally:
        G.ViewIndex.__init__ = orig_init


def test_backward_with_an_all_background_view_and_ragged_chunks(golden):
    """Edge cases of the per-view indices: a view none of whose pixels has a non-zero weight (0 entries, 0 rows: nothing to
    reduce, only pointers that must stay valid), next to ordinary views, and entry counts that are not multiples of the
    512-entry chunks. The deterministic gradient equals the float-atomic one to rounding and is repeatable bit for bit."""
    from nerfail_amd import GaussNet as G
    g = golden('g10_gauss_net')
    rs = np.random.RandomState(3)
    wi_np = np.concatenate([g['wi'], g['wi'][:1]], 0)
    wi_np[1, 0] = 0.0                                             # view 1: all background
    wi_np[2, 0][rs.uniform(size=wi_np[2, 0].shape) < 0.37] = 0.0  # view 2: ragged number of entries
    ori = T(np.concatenate([g['ori'], g['ori'][:1]], 0))
    Gr = T(np.concatenate([g['Gr'], g['Gr'][:1]], 0))
    wi, s0 = T(wi_np), T(g['s'])
    G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()
    vis = G.view_indices(wi, s0.numel() // 4)
    assert vis[1].n_entries == 0 and vis[1].n_rows == 0 and vis[2].n_entries % 512 != 0

    def grad(det):
        s = s0.clone().requires_grad_(True)
        x, xr = G.gauss_gather(s, wi, ori, 32.0, None, det)
        (xr * Gr).sum().backward()
        return s.grad
    a, b, ref = grad(True), grad(True), grad(False)
    assert torch.equal(a, b) and rel_err(N(a), N(ref)) < 1e-4


def test_backward_of_more_views_than_one_launch_holds():
    """nerfail_gauss_bwd_views reduces up to 16 views per launch; a batch of 19 takes two rounds, the second accumulating
    onto the first (fixed order: bitwise repeatable; equal to the float-atomic form to rounding)."""
    from nerfail_amd import GaussNet as G
    rs = np.random.RandomState(12)
    B, P, H, W = 19, 3, 12, 10
    s0 = T(rs.uniform(-40, 40, size=(P, H, W, 4)).astype(np.float32))
    ori = T(synth.disc_alpha_image(B, H, W, seed=13))
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, size=(B, H, W, 8)).astype(np.float32)
    wi, _ = G.create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    Gr = T(rs.normal(size=(B, H, W, 4)).astype(np.float32))

    def grad(det):
        s = s0.clone().requires_grad_(True)
        x, xr = G.gauss_gather(s, wi, ori, 24.0, None, det)
        ((xr * Gr).sum() + x.sum()).backward()
        return s.grad
    a, b, ref = grad(True), grad(True), grad(False)
    assert torch.equal(a, b) and rel_err(N(a), N(ref)) < 1e-4


def _toy_attack(seed=21, B=5, P=3, H=14, W=12):
    from nerfail_amd import GaussNet as G
    rs = np.random.RandomState(seed)
    s0 = rs.uniform(-40, 40, size=(P, H, W, 4)).astype(np.float32)
    s0[..., 3] = np.where(rs.uniform(size=(P, H, W)) < 0.8, 255.0, 0.0)
    ori_u8 = synth.disc_alpha_image(B, H, W, seed=seed + 1).astype(np.uint8)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    dist[:, :2] = 1.0                                                # some background rows (all weights 0)
    idx = rs.randint(0, P * H * W, size=(B, H, W, 8)).astype(np.float32)
    wi, _ = G.create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    torch.manual_seed(seed)
    victim = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(2), torch.nn.Flatten(),
                                 torch.nn.Linear(16, 8)).to(dev()).requires_grad_(False)
    return T(s0), wi, ori_u8, victim


@pytest.mark.parametrize('eps', [None, 24.0])
def test_rgb_only_attack_step_equals_the_full_gradient_path(eps):
    """The NeRFail-S step reads grad[..., :3] only (AS:357-392). Its fast path (no x tensor, alpha + 3-bit mask between forward
    and backward, gradient as [Ns,3] with the loss in the buffer's tail) must give the SAME BITS as the full autograd path:
    the rgb gradient, the loss, and the updated perturbation."""
    from nerfail_amd import GaussNet as G, attack as A
    s0, wi, ori_u8, victim = _toy_attack()
    ori = T(ori_u8.astype(np.float32))
    label = torch.tensor(3, device=dev())
    net = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=eps)
    g_full, loss_full, _ = A.perturbation_grad(net, s0, wi, ori, label)
    buf, _ = A.perturbation_grad_rgb(net, s0, wi, ori, label)
    Ns = s0.numel() // 4
    assert torch.equal(buf[:3 * Ns].view(Ns, 3), g_full.reshape(Ns, 4)[:, :3].contiguous())
    assert float(buf[3 * Ns]) == float(loss_full)
    assert float(g_full.abs().max()) > 0
    net.rgb_grad_only = False
    s_a, l_a = A.nerfail_s_step(net, s0, s0, wi, ori, label, 2.0, 32.0, False)
    net.rgb_grad_only = True
    s_b, l_b = A.nerfail_s_step(net, s0, s0, wi, ori, label, 2.0, 32.0, False)
    assert torch.equal(s_a, s_b) and float(l_a) == float(l_b)
    # uint8 images (what cv2.imread hands the reference's dataset, MyDataset.py:200) give the same bits as their float copy
    s_c, l_c = A.nerfail_s_step(net, s0, s0, wi, torch.from_numpy(ori_u8).to(dev()), label, 2.0, 32.0, False)
    assert torch.equal(s_c, s_b) and float(l_c) == float(l_b)


def test_views_resident_by_id_ignore_the_passed_host_tensors(tmp_path):
    """the reference's loop hands the step CPU tensors from a DataLoader every iteration
    (MyDataset.py:199-204, AS:304-317). Views named by id and resident on the device (load_view_maps / register_view /
    keep_views_resident) are taken from there: same bits as the passed-tensor path, and the passed tensors are not read."""
    from nerfail_amd import GaussNet as G, attack as A
    s0, wi, ori_u8, victim = _toy_attack(seed=33)
    label = torch.tensor(5, device=dev())
    Ns = s0.numel() // 4
    net = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=None)
    ref, ref_loss = A.nerfail_s_step(net, s0, s0, wi, torch.from_numpy(ori_u8).to(dev()), label, 2.0, 32.0, False)
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    # (a) maps on disk in the reference's layout -> load_view_maps, images registered by id
    mdir = str(tmp_path / 'index_and_weight')
    os.makedirs(mdir)
    for i in range(wi.shape[0]):
        torch.save(wi[i].cpu(), os.path.join(mdir, '%d.pth' % i))
    ids = G.load_view_maps(mdir, list(range(wi.shape[0])), Ns, images={i: ori_u8[i] for i in range(wi.shape[0])})
    junk_wi = torch.zeros(tuple(wi.shape))                       # CPU tensors of the right shape and WRONG content
    junk_ori = torch.zeros(ori_u8.shape, dtype=torch.uint8)
    got, got_loss = A.nerfail_s_step(net, s0, s0, junk_wi, junk_ori, label, 2.0, 32.0, False, view_ids=ids)
    assert torch.equal(got, ref) and float(got_loss) == float(ref_loss)
    got2, _ = A.nerfail_s_step(net, s0, s0, None, None, label, 2.0, 32.0, False, view_ids=ids)      # nothing passed at all
    assert torch.equal(got2, ref)
    # shuffled composition (the DataLoader shuffles): the right views are found by id
    perm = [3, 0, 4, 1, 2]
    ref_p, _ = A.nerfail_s_step(net, s0, s0, wi[perm].contiguous(), torch.from_numpy(ori_u8[perm]).to(dev()), label, 2.0, 32.0, False)
    got_p, _ = A.nerfail_s_step(net, s0, s0, junk_wi, junk_ori, label, 2.0, 32.0, False, view_ids=[ids[k] for k in perm])
    assert torch.equal(got_p, ref_p)
    # the full forward() of the module (reference signature) finds them too
    x, xr, cla, o, ocla = net(s0, junk_wi, junk_ori, view_ids=ids)
    x_r, xr_r, cla_r, o_r, _ = net(s0, wi, torch.from_numpy(ori_u8).to(dev()))
    assert torch.equal(x, x_r) and torch.equal(xr, xr_r) and torch.equal(o, o_r) and torch.equal(cla, cla_r)
    # (b) keep_views_resident: the first sight of an id uploads and keeps the view, later calls do not read the host tensors
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    net.keep_views_resident = True
    names = [('scene', 'test', i) for i in range(wi.shape[0])]
    first, _ = A.nerfail_s_step(net, s0, s0, wi.cpu(), torch.from_numpy(ori_u8), label, 2.0, 32.0, False, view_ids=names)
    later, _ = A.nerfail_s_step(net, s0, s0, junk_wi, junk_ori, label, 2.0, 32.0, False, view_ids=names)
    assert torch.equal(first, ref) and torch.equal(later, ref)


def test_fingerprint_mixes_every_word_with_its_position():
    """the content key of a map is two sums of murmur-finalised (word, position) hashes - restated here in
    numpy -, so swapping two words, moving a value or +d / -d at two positions all change it (the round-2 checksum, a plain
    sum + a sum weighted by position mod 65521, saw none of these at positions of equal weight)."""
    from nerfail_amd import GaussNet as G
    rs = np.random.RandomState(3)
    base = rs.randint(0, 2 ** 31, size=(1, 2, 40, 41, 8)).astype(np.int32)          # 26 240 words per map, 16-byte aligned

    def fmix(h):
        h = h.astype(np.uint64)
        for sh, mul in ((16, 0x85ebca6b), (13, 0xc2b2ae35)):
            h ^= h >> np.uint64(sh)
            h = (h * np.uint64(mul)) & np.uint64(0xffffffff)
        return h ^ (h >> np.uint64(16))

    def ref(words):
        w = words.reshape(-1).view(np.uint32).astype(np.uint64)
        pos = np.arange(w.size, dtype=np.uint64)
        m32 = np.uint64(0xffffffff)
        a = fmix(w ^ fmix((pos * np.uint64(0x9e3779b1) + np.uint64(1)) & m32))
        b = fmix(((w + np.uint64(0x7f4a7c15)) & m32) ^ fmix((pos * np.uint64(0x85ebca77) + np.uint64(0x165667b1)) & m32))
        return int(a.sum() & np.uint64(0xffffffffffffffff)), int(b.sum() & np.uint64(0xffffffffffffffff))

    def fp(words):
        k = G.fingerprints(T(words.view(np.float32)))[0]
        return k[1] & 0xffffffffffffffff, k[2] & 0xffffffffffffffff

    assert fp(base) == ref(base)
    flat = base.reshape(-1)
    swapped = flat.copy(); swapped[[5, 5 + 65521 % flat.size]] = swapped[[5 + 65521 % flat.size, 5]]
    plus_minus = flat.copy(); plus_minus[7] += 9; plus_minus[8] -= 9
    moved = np.roll(flat, 4)
    seen = {fp(base)}
    for other in (swapped, plus_minus, moved):
        k = fp(other.reshape(base.shape))
        assert k == ref(other) and k not in seen
        seen.add(k)


def test_resident_views_are_checked_against_passed_device_maps_and_logit_cache_follows_the_image(tmp_path):
    """(i) a view that is resident under an id wins over the tensor passed under that id - but ONCE per id a
    passed DEVICE map is compared with the resident one, so that a reused id / regenerated map raises instead of silently
    differentiating through another view; (ii) register_view of a new image under an id invalidates the cached original-image
    logits of that id; (iii) MyDataset view ids carry the files' (mtime, size)."""
    from nerfail_amd import GaussNet as G, attack as A
    from nerfail_amd.MyDataset import gauss_dataset
    s0, wi, ori_u8, victim = _toy_attack(seed=55)
    label = torch.tensor(2, device=dev())
    Ns = s0.numel() // 4
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear(); G._VIEW_PASSED_OK.clear()
    ids = [('scene', 'split', i) for i in range(wi.shape[0])]
    for i, vid in enumerate(ids):
        G.register_view(vid, Ns, weight_and_index=wi[i], ori_img=ori_u8[i])
    net = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=None)
    net.cache_ori_cla = True
    ori_dev = torch.from_numpy(ori_u8).to(dev())
    ref, _ = A.nerfail_s_step(net, s0, s0, wi, ori_dev, label, 2.0, 32.0, False, view_ids=ids)      # same maps passed: accepted
    assert all(G._view_key(v, Ns) in G._VIEW_PASSED_OK for v in ids)
    # (i) another scene's maps under the same ids, as a device tensor: caught on first sight
    G._VIEW_PASSED_OK.clear()
    with pytest.raises(ValueError, match='DIFFERENT map'):
        A.nerfail_s_step(net, s0, s0, wi.flip(0).contiguous(), ori_dev, label, 2.0, 32.0, False, view_ids=ids)
    # (ii) the image of view 0 is replaced: the step must see the new image's logits (cache keyed with the image epoch)
    _, cla0, ori_cla0, _, _ = net.attack_forward(s0, None, None, view_ids=ids)
    new_img = ori_u8[0].copy()
    new_img[..., :3] = 255 - new_img[..., :3]
    G.register_view(ids[0], Ns, ori_img=new_img)
    _, cla1, ori_cla1, _, _ = net.attack_forward(s0, None, None, view_ids=ids)
    fresh = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=None)
    _, _, ori_cla_ref, _, _ = fresh.attack_forward(s0, None, None, view_ids=ids)
    assert not torch.equal(ori_cla1[0], ori_cla0[0]) and torch.equal(ori_cla1, ori_cla_ref)
    # (iii) dataset ids change with the file
    mp, ip = str(tmp_path / '0.pth'), str(tmp_path / '0.png')
    torch.save(wi[0].cpu(), mp)
    from PIL import Image
    Image.fromarray(ori_u8[0][..., [2, 1, 0, 3]], 'RGBA').save(ip)
    a = gauss_dataset([mp], [ip], [''], [''], dev(), Ns=Ns).view_id(0)
    os.utime(mp, ns=(5, 5))
    b = gauss_dataset([mp], [ip], [''], [''], dev(), Ns=Ns).view_id(0)
    assert a != b and a[0] == b[0] == os.path.abspath(mp)
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear(); G._VIEW_PASSED_OK.clear()


def test_view_ids_are_checked_and_stale_sidecars_are_rebuilt(tmp_path):
    """ADVICE r2 (medium x2): (i) an id reused for a view of another resolution raises instead of reading out of bounds, an
    id whose registered index belongs to another map is caught by the fingerprint on first use, register_view_index
    refuses to overwrite silently; (ii) a <i>.idx.pth sidecar built from another version of <i>.pth is rebuilt."""
    from nerfail_amd import GaussNet as G
    s0, wi, ori_u8, _ = _toy_attack(seed=44)
    ori = T(ori_u8.astype(np.float32))
    Ns = s0.numel() // 4
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    G.view_indices(wi[:2], Ns, [0, 1])
    small = wi[:2, :, :7].contiguous()                             # "test view 0" at another resolution under the same bare id
    with pytest.raises(ValueError):
        G.view_indices(small, Ns, [0, 1])
    # an index of ANOTHER map registered under an id: caught when the id is first used with a map at hand
    other = G.ViewIndex(wi[3], Ns)
    other.verified = False
    G.register_view_index('v9', other, Ns)
    with pytest.raises(ValueError):
        G.view_indices(wi[2:3], Ns, ['v9'])
    with pytest.raises(ValueError):
        G.register_view_index('v9', G.ViewIndex(wi[4, :, :7].contiguous(), Ns), Ns)       # silent overwrite refused
    # stale sidecar
    mdir = str(tmp_path / 'maps')
    os.makedirs(mdir)
    torch.save(wi[0].cpu(), os.path.join(mdir, '0.pth'))
    ids = G.load_view_indices(mdir, [0], Ns)
    a = G._VIEW_CACHE[G._view_key(ids[0], Ns)]
    torch.save(wi[1].cpu(), os.path.join(mdir, '0.pth'))           # the map is regenerated (another c / NeRF / resolution)
    os.utime(os.path.join(mdir, '0.pth'), ns=(1, 1))
    ids = G.load_view_indices(mdir, [0], Ns)
    b = G._VIEW_CACHE[G._view_key(ids[0], Ns)]
    assert tuple(a.fp) != tuple(b.fp) and tuple(b.fp) == tuple(G.fingerprints(wi[1:2])[0])
    s = s0.clone().requires_grad_(True)
    x, xr = G.gauss_gather(s, wi[1:2], ori[1:2], None, None, True, ids)
    xr.sum().backward()
    s2 = s0.clone().requires_grad_(True)
    x2, xr2 = G.gauss_gather(s2, wi[1:2], ori[1:2], None, None, True)
    xr2.sum().backward()
    assert torch.equal(s.grad, s2.grad)


def test_dataset_mirror_feeds_the_step_from_resident_views(tmp_path):
    """nerfail_amd.MyDataset.gauss_dataset (mirror of MyDataset.py:187-204) behind a DataLoader(num_workers=0), as AS:222-231
    / AS:304-317 drive the step: with its collate_views the batch is a list of device-resident views (read once), with the
    default collate a stacked device tensor as in the reference - both give the bits of the plain tensor path."""
    from PIL import Image
    from nerfail_amd import GaussNet as G, attack as A
    from nerfail_amd.MyDataset import gauss_dataset
    s0, wi, ori_u8, victim = _toy_attack(seed=55, B=4)
    label = torch.tensor(2, device=dev())
    Ns = s0.numel() // 4
    net = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=None)
    ref, ref_loss = A.nerfail_s_step(net, s0, s0, wi, torch.from_numpy(ori_u8).to(dev()), label, 2.0, 32.0, False)
    maps, pngs = [], []
    for b in range(4):
        maps.append(str(tmp_path / ('%d.pth' % b)))
        pngs.append(str(tmp_path / ('%d.png' % b)))
        torch.save(wi[b].cpu(), maps[-1])
        Image.fromarray(ori_u8[b][..., [2, 1, 0, 3]], 'RGBA').save(pngs[-1])          # BGRA array -> RGBA file (cv2 reads BGRA back)
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    for own in (True, False):
        ds = gauss_dataset(maps, pngs, [''] * 4, [''] * 4, dev(), Ns=Ns)
        loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0, collate_fn=ds.collate_views if own else None)
        for epoch in range(2):                                   # second epoch: everything comes from the device
            for idx, ori_b, wi_b, _, _ in loader:
                got, got_loss = A.nerfail_s_step(net, s0, s0, wi_b, ori_b, label, 2.0, 32.0, False)
                assert torch.equal(got, ref) and float(got_loss) == float(ref_loss), (own, epoch)
    assert len(G._VIEW_MAPS) == 4 and len(G._VIEW_ORI) == 4


def test_attack_step_reuses_cached_original_logits():
    """cache_ori_cla on the step's fast path: the original images' logits are computed once per image tensor (device tensor
    identity) or per set of view ids, not in every step (GN:157 recomputes them every forward)."""
    from nerfail_amd import GaussNet as G, attack as A
    s0, wi, ori_u8, victim = _toy_attack(seed=66, B=3)
    calls = []

    class Counting(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            calls.append(x.shape[0])
            return self.m(x)
    net = G.gauss_net(dev(), 0.02, Counting(victim), 'my_model', epsilon=None)
    label = torch.tensor(1, device=dev())
    ori_d = torch.from_numpy(ori_u8).to(dev())
    ref, _ = A.nerfail_s_step(net, s0, s0, wi, ori_d, label)
    assert len(calls) == 2                                       # perturbed + original
    net.cache_ori_cla = True
    calls.clear()
    a, _ = A.nerfail_s_step(net, s0, s0, wi, ori_d, label)
    b, _ = A.nerfail_s_step(net, s0, s0, wi, ori_d, label)
    assert len(calls) == 3 and torch.equal(a, ref) and torch.equal(b, ref)      # 2 + 1: the second step reuses the logits
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    net.keep_views_resident = True
    calls.clear()
    ids = [('s', i) for i in range(3)]
    c, _ = A.nerfail_s_step(net, s0, s0, wi.cpu(), torch.from_numpy(ori_u8), label, view_ids=ids)
    d, _ = A.nerfail_s_step(net, s0, s0, None, None, label, view_ids=ids)
    assert len(calls) == 3 and torch.equal(c, ref) and torch.equal(d, ref)


def test_original_logits_are_cached_by_default_for_named_views():
    """cache_ori_cla = None (the default) keeps the unperturbed images' logits per set of view ids while the
    classifier is a frozen pure function (every module in eval() mode, no parameter requiring grad: AS:281-287) - the loop of
    INTEGRATION.md section 1 pays ONE classifier forward per step. Identical bits to the recomputing path; invalidated by
    train() mode, by a parameter that requires grad, by any write to a weight or buffer, and by a replaced view image; views
    without ids are never cached by default."""
    from nerfail_amd import GaussNet as G, attack as A
    s0, wi, ori_u8, victim = _toy_attack(seed=77, B=3)
    calls = []

    class Counting(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            calls.append(x.shape[0])
            return self.m(x)
    model = Counting(victim)
    model.train(False)                                            # AS:281
    net = G.gauss_net(dev(), 0.02, model, 'my_model', epsilon=None)
    assert net.cache_ori_cla is None
    label = torch.tensor(1, device=dev())
    ori_d = torch.from_numpy(ori_u8).to(dev())
    ids = [('r6', i) for i in range(3)]
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()
    net.cache_ori_cla = False
    ref, ref_loss = A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    ref5 = net(s0, wi, ori_d, view_ids=ids)
    net.cache_ori_cla = None
    calls.clear()
    a, la = A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)            # computes + stores
    b, lb = A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)            # reuses
    assert calls == [3, 3, 3] and torch.equal(a, ref) and torch.equal(b, ref) and float(la) == float(lb) == float(ref_loss)
    calls.clear()
    got5 = net(s0, wi, ori_d, view_ids=ids)                                          # the reference-shaped 5-tuple forward too
    assert calls == [3] and all(torch.equal(x, y) for x, y in zip(got5, ref5))
    got5[4].add_(1000.)                                                              # the caller owns what it got (deepfool.py:54-57 writes
    again5 = net(s0, wi, ori_d, view_ids=ids)                                        # into logits in place): the cache keeps its own copy
    assert torch.equal(again5[4], ref5[4])
    calls.clear()
    calls.clear()
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label)                                  # no ids: never cached by default
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label)
    assert calls == [3, 3, 3, 3]
    # a subset / another order of the same views is another key
    calls.clear()
    A.nerfail_s_step(net, s0, s0, wi[:2].contiguous(), ori_d[:2].contiguous(), label, view_ids=ids[:2])
    assert calls == [2, 2]
    # train() mode: not a pure function (dropout, batch-norm statistics) -> recomputed every step
    model.train(True)
    calls.clear()
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    assert calls == [3, 3]
    model.train(False)
    calls.clear()
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    assert calls == [3]                                                               # still valid: nothing was written
    # a weight written in place (an optimizer step, load_state_dict): every stored logit is stale
    with torch.no_grad():
        victim[4].weight.mul_(1.5)
    calls.clear()
    c, _ = A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    assert calls == [3, 3]
    net.cache_ori_cla = False
    c_ref, _ = A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    assert torch.equal(c, c_ref)
    net.cache_ori_cla = None
    # a parameter that requires grad: something is being learned through the classifier -> no caching
    victim[4].weight.requires_grad_(True)
    calls.clear()
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    A.nerfail_s_step(net, s0, s0, wi, ori_d, label, view_ids=ids)
    assert calls == [3, 3, 3, 3]
    victim[4].weight.requires_grad_(False)
    G._VIEW_CACHE.clear(); G._VIEW_MAPS.clear(); G._VIEW_ORI.clear()


@pytest.mark.parametrize('layout', ['random', 'few_long_rows', 'one_row', 'runs'])
@pytest.mark.parametrize('targeted,eps', [(False, None), (True, 24.0)])
def test_fused_sign_step_equals_the_separate_kernels(layout, targeted, eps):
    """Round 6. On one rank nerfail_s_step runs K11 with the sign step as the epilogue of its last launch
    (nerfail_gauss_bwd_views_rgb_step). Every bit stays as it was: the fused step == rgb gradient + igsm_step_rgb == the
    four-channel autograd path, on index layouts that make rows cross one, several and all 512-entry chunks."""
    from nerfail_amd import GaussNet as G, attack as A
    rs = np.random.RandomState(31)
    P_, B, H, W = 3, 3, 40, 36                                    # 1 440 pixels x 8 = 11 520 entries per view: 23 chunks
    Ns = P_ * H * W
    s0 = rs.uniform(-40, 40, size=(P_, H, W, 4)).astype(np.float32)
    s0[..., 3] = np.where(rs.uniform(size=(P_, H, W)) < 0.8, 255.0, 0.0)
    ori_u8 = synth.disc_alpha_image(B, H, W, seed=32).astype(np.uint8)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    if layout == 'random':
        idx = rs.randint(0, Ns, size=(B, H, W, 8))
    elif layout == 'few_long_rows':                               # 7 rows of ~1 600 entries each: every row spans 3-4 chunks
        idx = rs.randint(0, 7, size=(B, H, W, 8)) * 601
    elif layout == 'one_row':                                     # ONE row holds all entries of a view: head / middle / tail records only
        idx = np.full((B, H, W, 8), 1234)
    else:                                                         # runs of 300-700 entries: most rows cross exactly one boundary
        idx = (np.arange(B * H * W * 8).reshape(B, H, W, 8) // rs.randint(300, 700)) % Ns
    wi, _ = G.create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx.astype(np.float32)], 1)))
    torch.manual_seed(5)
    victim = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(2), torch.nn.Flatten(),
                                 torch.nn.Linear(16, 8)).to(dev()).requires_grad_(False)
    with torch.no_grad():
        victim[0].weight.mul_(0.05)                               # logits O(1): a real gradient
    net = G.gauss_net(dev(), 0.02, victim, 'my_model', epsilon=eps)
    label = torch.tensor(3, device=dev())
    s0t, ori = T(s0), torch.from_numpy(ori_u8).to(dev())
    s_init = T(s0 * 0.5)
    g_full, loss_full, _ = A.perturbation_grad(net, s0t, wi, ori.float(), label)              # four channels, combine kernel
    buf, _ = A.perturbation_grad_rgb(net, s0t, wi, ori, label)                                # rgb form
    assert float(g_full.abs().max()) > 1e-8
    assert torch.equal(buf[:3 * Ns].view(Ns, 3), g_full.reshape(Ns, 4)[:, :3].contiguous()), layout
    ref = A.igsm_step_rgb(s0t, buf, s_init, 2.0, 32.0, targeted)
    fused, loss = A.perturbation_step_rgb(net, s0t, s_init, wi, ori, label, 2.0, 32.0, targeted)
    assert torch.equal(fused.view(-1), ref.view(-1)) and float(loss) == float(loss_full), layout
    # the product's step function: fused on one rank, switchable
    a1, l1 = A.nerfail_s_step(net, s0t, s_init, wi, ori, label, 2.0, 32.0, targeted)
    net.fused_sign_step = False
    a2, l2 = A.nerfail_s_step(net, s0t, s_init, wi, ori, label, 2.0, 32.0, targeted)
    assert torch.equal(a1, a2) and torch.equal(a1.view(-1), ref.view(-1)) and float(l1) == float(l2)
    # the gradient on request, next to the step
    xr, cla, _, views, aux = net.attack_forward(s0t, wi, ori, None)
    torch.nn.functional.cross_entropy(cla, label.broadcast_to([B]), reduction='sum').div(B).backward()
    gout = torch.empty((3 * Ns,), device=dev())
    again = G.hot_backward_rgb_step(aux, xr.grad, views, s0t, s_init, 2.0, 32.0, targeted, grad_out=gout)
    assert torch.equal(again.view(-1), ref.view(-1)) and torch.equal(gout, buf[:3 * Ns])
    # repeatable
    for _ in range(2):
        f2, _ = A.perturbation_step_rgb(net, s0t, s_init, wi, ori, label, 2.0, 32.0, targeted)
        assert torch.equal(f2, fused)
