"""-m gpu: exact 8-NN build through the C ABI, BIT-EXACT against oracle/knn.py (indices and distances)."""
import numpy as np
import torch
import pytest

import synth
from hiputil import T, N
from oracle import knn as OK

pytestmark = pytest.mark.gpu


def _cases(golden):
    g = golden('g8_knn')
    yield g['a_Q'], g['a_S']
    yield (synth.sphere_shell_points(24 * 24, seed=int(g['b_Q_seed'])).reshape(24, 24, 3),
           synth.sphere_shell_points(20000, seed=int(g['b_S_seed'])))


def test_knn_bit_exact_vs_oracle(golden):
    from nerfail_amd.create_index_and_dist import knn8, index_and_dist
    for Q, S in _cases(golden):
        od, oi = OK.knn8(Q.reshape(-1, 3), S)
        for method in ('brute', 'grid'):
            d, i = knn8(T(Q), T(S), want_int=True, method=method)
            assert np.array_equal(N(i).reshape(-1, 8), oi), method
            assert np.array_equal(N(d).reshape(-1, 8), od), method
        out = N(index_and_dist(T(Q), T(S)))
        assert out.shape == (2,) + Q.shape[:2] + (8,) and out.dtype == np.float32
        assert np.array_equal(out, OK.index_and_dist(Q, S))


def test_knn_ragged_sizes():
    from nerfail_amd.create_index_and_dist import knn8
    for nq, npnt in ((1, 8), (513, 2049), (7, 4096), (1000, 9)):
        S = synth.sphere_shell_points(npnt, seed=nq)
        Q = synth.sphere_shell_points(nq, seed=npnt + 1)
        S[npnt // 2] = S[0]                                   # a tie pair
        od, oi = OK.knn8(Q, S)
        for method in ('brute', 'grid'):
            d, i = knn8(T(Q), T(S), want_int=True, method=method)
            assert np.array_equal(N(i), oi) and np.array_equal(N(d), od), method


def test_knn_rejects_too_few_points():
    from nerfail_amd.create_index_and_dist import knn8
    from nerfail_amd._lib import NerfailError
    with pytest.raises(NerfailError):
        knn8(T(np.zeros((4, 3), np.float32)), T(np.zeros((7, 3), np.float32)))


def test_knn_grid_adversarial_inputs_match_brute_force():
    """Grid search == brute force bit for bit on inputs that stress the termination bound and the cell logic:
    queries far outside the bounding box, a degenerate (planar / collinear) point set, heavy duplicates, one dense
    cluster plus far outliers, and the full-size shape (1.92 M points) on a query subset."""
    from nerfail_amd.create_index_and_dist import knn8
    rs = np.random.RandomState(7)
    cases = []
    S = synth.sphere_shell_points(50000, seed=1)
    cases.append((np.concatenate([rs.uniform(-5, 5, (300, 3)), rs.uniform(-1, 1, (300, 3))]).astype(np.float32), S))
    plane = rs.uniform(-1, 1, (30000, 3)).astype(np.float32)
    plane[:, 2] = 0.25
    cases.append((rs.uniform(-1, 1, (500, 3)).astype(np.float32), plane))
    line = np.zeros((20000, 3), np.float32)
    line[:, 0] = rs.uniform(-1, 1, 20000)
    cases.append((rs.uniform(-1, 1, (300, 3)).astype(np.float32), line))
    dup = np.repeat(rs.uniform(-1, 1, (700, 3)).astype(np.float32), 12, axis=0)          # every point 12x
    cases.append((dup[::7].copy(), dup))
    clus = np.concatenate([rs.normal(scale=1e-3, size=(40000, 3)), rs.uniform(-50, 50, (64, 3))]).astype(np.float32)
    cases.append((np.concatenate([rs.normal(scale=1e-3, size=(400, 3)), rs.uniform(-50, 50, (100, 3))]).astype(np.float32), clus))
    for Q, P in cases:
        db, ib = knn8(T(Q), T(P), want_int=True, method='brute')
        dg, ig = knn8(T(Q), T(P), want_int=True, method='grid')
        assert torch_equal(ib, ig) and torch_equal(db, dg)
    # full-size point set (3 x 800 x 800), 4096 queries
    P = synth.sphere_shell_points(3 * 800 * 800, seed=0)
    Q = synth.sphere_shell_points(4096, seed=1)
    db, ib = knn8(T(Q), T(P), want_int=True, method='brute')
    dg, ig = knn8(T(Q), T(P), want_int=True, method='grid')
    assert torch_equal(ib, ig) and torch_equal(db, dg)


def torch_equal(a, b):
    import torch
    return torch.equal(a, b)


def test_knn_pipeline_vs_reference_chain(golden):
    """Fixture g17: K8 -> K9 -> K10 (x) against the reference's own chain (torch.cdist matmul path + create_gauss_w +
    gauss_net) in the real regime (neighbour spacing << c = 0.02, exact hits present). K8 is bit-exact vs the oracle; the
    deviation of x from the reference PIPELINE is measured here and bounded (DESIGN.md section 2 records it)."""
    from test_oracle_knn import agreement_with_reference, pipeline_deviation
    from nerfail_amd.create_index_and_dist import index_and_dist
    from nerfail_amd.GaussNet import create_gauss_w, gauss_gather
    from hiputil import dev
    g = golden('g17_knn_pipeline')
    dai = index_and_dist(T(g['Q']), T(g['S']), method='grid')
    assert np.array_equal(N(dai), OK.index_and_dist(g['Q'], g['S']))
    assert np.array_equal(N(index_and_dist(T(g['Q']), T(g['S']), method='brute')), N(dai))
    wi, _ = create_gauss_w(dev(), 0.02)(dai.unsqueeze(0))
    x, _ = gauss_gather(T(g['s']), wi, T(g['ori']), None)
    frac, worst = pipeline_deviation(N(x), g['ref_x'])
    ref = g['ref_dist_and_index']
    m = agreement_with_reference(N(dai)[0].reshape(-1, 8), N(dai)[1].reshape(-1, 8).astype(np.int64),
                                 ref[0].reshape(-1, 8), ref[1].reshape(-1, 8).astype(np.int64))
    print('HIP K8->K9->K10 vs reference chain: ordered %.4f sets %.4f, pixels with |dx| > 1e-4 max|x|: %.4f (worst %.3e)'
          % (m['ordered'], m['sets'], frac, worst))
    assert m['sets'] > 0.97 and frac < 0.03 and worst < 0.05
    # fed with the REFERENCE's own 8-NN output, K9 + K10 reproduce the reference's x to 1e-5: the deviation is all K8's
    wi_r, _ = create_gauss_w(dev(), 0.02)(T(ref[None]))
    x_r, _ = gauss_gather(T(g['s']), wi_r, T(g['ori']), None)
    frac_r, worst_r = pipeline_deviation(N(x_r), g['ref_x'])
    assert frac_r == 0.0 and worst_r < 1e-5, (frac_r, worst_r)


def test_knn_far_queries_on_view_geometry():
    """Rendered-view geometry (synth.sphere_view_points: ~40 % surface pixels, ~60 % background pixels whose points sit
    on the near plane, one to two units from every point of the set): the background queries leave the fine shell search
    and finish on the coarse grid with box pruning. Grid result == brute force == oracle, bit for bit."""
    from nerfail_amd.create_index_and_dist import index_and_dist
    H = W = 96
    S = np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)
    for th in (-171., 33.):
        Q = synth.sphere_view_points(H, W, th)
        grid = N(index_and_dist(T(Q), T(S), method='grid'))
        brute = N(index_and_dist(T(Q), T(S), method='brute'))
        assert np.array_equal(grid, brute), th
        sel = np.concatenate([np.arange(0, 40), np.arange(H * W // 2, H * W // 2 + 40)])          # background + surface rows
        od, oi = OK.knn8(Q.reshape(-1, 3)[sel], S)
        assert np.array_equal(grid[0].reshape(-1, 8)[sel], od) and np.array_equal(grid[1].reshape(-1, 8)[sel], oi.astype(np.float32))
        far = grid[0][..., 0] > 0.5
        assert 0.2 < far.mean() < 0.8                                # the far-query path really ran (0.39 and 0.26 here)


def test_knn_view_tiles_cover_odd_sizes_and_equal_the_flat_search():
    """nerfail_knn8_grid_search_view (a wave = an 8 x 8 pixel tile) on views whose sides are not multiples of 8 - edge tiles
    with repeated lanes - equals the flat search and brute force on the same points, bit for bit, every pixel written."""
    from nerfail_amd.create_index_and_dist import knn8
    for H, W in ((37, 53), (8, 8), (9, 130)):
        S = np.stack([synth.sphere_view_points(64, 64, th) for th in (-120., 0., 120.)]).reshape(-1, 3)
        Q = synth.sphere_view_points(max(H, W), max(H, W), 45.)[:H, :W].copy()
        dv, iv = knn8(T(Q), T(S), want_int=True, method='grid')                      # [H,W,3] -> the tiled entry point
        df, i_f = knn8(T(Q.reshape(-1, 3)), T(S), want_int=True, method='grid')      # flat: 64 consecutive queries per wave
        db, ib = knn8(T(Q.reshape(-1, 3)), T(S), want_int=True, method='brute')
        assert dv.shape == (H, W, 8)
        assert torch_equal(dv.reshape(-1, 8), df) and torch_equal(iv.reshape(-1, 8), i_f), (H, W)
        assert torch_equal(df, db) and torch_equal(i_f, ib), (H, W)


def test_knn_full_size_view_geometry_grid_equals_brute_force():
    """BASELINE-size map build on the geometry of a real view: all 640 000 queries of an 800 x 800 view (31 % of them
    background pixels on the near plane) against the 1.92 M points of 3 base views - the grid search (near path + coarse
    far path) returns exactly what the brute-force scan returns, distances and indices, every query."""
    from nerfail_amd.create_index_and_dist import index_and_dist
    H = W = 800
    S = T(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3))
    Q = T(synth.sphere_view_points(H, W, 45.).reshape(H, W, 3))
    grid = index_and_dist(Q, S, method='grid')
    brute = index_and_dist(Q, S, method='brute')
    assert torch.equal(grid, brute)
    far = (grid[0][..., 0] > 0.5).float().mean().item()
    assert 0.2 < far < 0.8
