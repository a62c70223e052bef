"""-m gpu: exact 8-NN build through the C ABI, BIT-EXACT against oracle/knn.py (indices and distances)."""
import numpy as np
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
        d, i = knn8(T(Q), T(S), want_int=True)
        od, oi = OK.knn8(Q.reshape(-1, 3), S)
        assert np.array_equal(N(i).reshape(-1, 8), oi)
        assert np.array_equal(N(d).reshape(-1, 8), od)
        out = N(index_and_dist(T(Q), T(S)))
        assert out.shape == (2,) + Q.shape[:2] + (8,) and out.dtype == np.float32
        assert np.array_equal(out, OK.index_and_dist(Q, S))


def test_knn_ragged_sizes():
    from nerfail_amd.create_index_and_dist import knn8
    for nq, npnt in ((1, 8), (513, 2049), (7, 4096), (1000, 9)):
        S = synth.sphere_shell_points(npnt, seed=nq)
        Q = synth.sphere_shell_points(nq, seed=npnt + 1)
        S[npnt // 2] = S[0]                                   # a tie pair
        d, i = knn8(T(Q), T(S), want_int=True)
        od, oi = OK.knn8(Q, S)
        assert np.array_equal(N(i), oi) and np.array_equal(N(d), od)


def test_knn_rejects_too_few_points():
    from nerfail_amd.create_index_and_dist import knn8
    from nerfail_amd._lib import NerfailError
    with pytest.raises(NerfailError):
        knn8(T(np.zeros((4, 3), np.float32)), T(np.zeros((7, 3), np.float32)))
