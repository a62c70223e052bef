"""Pin oracle/knn.py (the DEFINITION of the bit-exact 8-NN) against fixture g8.

Two written-down rules (SURVEY.md section 7 "8-NN bit-exact"):
 (i)  vs exact float64 ordering: index SETS must agree wherever the float64 gap between the 8th
      and 9th neighbour exceeds what float32 direct-difference arithmetic can resolve;
 (ii) vs the reference procedure's own output (torch.cdist matmul path + unstable sort, re-issued
      by make_golden.py): ordered indices must agree wherever the reference's own consecutive
      distances are separated by more than its error bound (|a|^2+|b|^2-2ab cancellation:
      ~1e-6 absolute in d^2 for unit-sphere points => ~ 1e-6/(2d) in d), and its distances must
      agree to that same bound.
"""
import numpy as np

import synth
from oracle import knn as O


def _case(g, tag):
    if tag == 'a':
        return g['a_Q'], g['a_S']
    S = synth.sphere_shell_points(20000, seed=int(g['b_S_seed']))
    Q = synth.sphere_shell_points(24 * 24, seed=int(g['b_Q_seed'])).reshape(24, 24, 3)
    return Q, S


def test_knn_against_exact_float64(golden):
    g = golden('g8_knn')
    for tag in ('a', 'b'):
        Q, S = _case(g, tag)
        d, i = O.knn8(Q.reshape(-1, 3), S)
        ex_i, ex_d9 = g[tag + '_exact64_idx'], g[tag + '_exact64_d9']
        gap = ex_d9[:, 8] - ex_d9[:, 7]
        clear = gap > 1e-6 * np.maximum(ex_d9[:, 8], 1e-3)
        assert clear.mean() > 0.95
        same_set = np.array([set(a) == set(b) for a, b in zip(i, ex_i)])
        assert same_set[clear].all()
        assert np.abs(d - ex_d9[:, :8]).max() < 1e-6
        assert (np.diff(d, axis=-1) >= 0).all()


def test_knn_ties_and_exact_hits(golden):
    g = golden('g8_knn')
    Q, S = _case(g, 'a')
    d, i = O.knn8(Q.reshape(-1, 3), S)
    assert (d[:8, 0] == 0).all() and (i[:8, 0] == np.arange(8)).all()     # exact hits -> distance 0
    # duplicate points 100/101: equal d2, ascending index order
    q = S[100][None]
    d1, i1 = O.knn8(q, S)
    assert list(i1[0, :2]) == [100, 101] and d1[0, 0] == 0 and d1[0, 1] == 0


def test_knn_against_reference_procedure_where_separated(golden):
    g = golden('g8_knn')
    for tag in ('a', 'b'):
        Q, S = _case(g, tag)
        out = O.index_and_dist(Q, S)
        ref = g[tag + '_ref']
        assert out.shape == ref.shape == (2,) + Q.shape[:2] + (8,)
        rd, ri = ref[0].reshape(-1, 8), ref[1].reshape(-1, 8).astype(np.int64)
        d, i = out[0].reshape(-1, 8), out[1].reshape(-1, 8).astype(np.int64)
        bound = 2e-6 / np.maximum(2 * rd, 1e-3)              # cdist matmul-path error in d
        sep_prev = np.concatenate([np.full((rd.shape[0], 1), np.inf), np.diff(rd, axis=-1)], -1)
        sep_next = np.concatenate([np.diff(rd, axis=-1), np.full((rd.shape[0], 1), np.inf)], -1)
        # the 8th entry also needs clearance from the (unstored) 9th: use the oracle's own 9th
        well = (sep_prev > 2 * bound) & (sep_next > 2 * bound)
        well[:, 7] = False
        assert well.mean() > 0.5
        assert (i[well] == ri[well]).all()
        assert (np.abs(d - rd)[well] <= bound[well]).all()
