"""Pin oracle/gauss.py against golden vectors produced by the reference (g9, g10, g11)."""
import numpy as np

from conftest import rel_err
from oracle import gauss as O


def test_create_gauss_w_matches_reference(golden):
    g = golden('g9_gauss_w')
    i_w, dist = O.create_gauss_w(g['dist_and_index'], c=0.02)
    assert rel_err(i_w[:, 0], g['i_w'][:, 0]) < 1e-5
    assert np.array_equal(i_w[:, 1], g['i_w'][:, 1])          # indices pass through untouched
    assert np.array_equal(dist, g['dist'])
    assert (i_w[0, 0, 0, 0] == 0).all()                       # sum g underflow -> w = 0 branch (GN:181)


def test_gauss_forward_matches_reference(golden):
    g = golden('g10_gauss_net')
    for tag, eps in (('epsNone_', None), ('eps32_', 32.0)):
        x, x_rgba, (emax, emin) = O.gauss_forward(g['s'], g['wi'], g['ori'], eps)
        assert rel_err(x, g[tag + 'x']) < 1e-5
        assert rel_err(x_rgba, g[tag + 'x_rgba']) < 1e-5
        assert abs(emax - float(g[tag + 'eps3d_max'])) < 1e-3 * abs(emax)
        assert abs(emin - float(g[tag + 'eps3d_min'])) < 1e-3 * abs(emin)


def test_gauss_backward_matches_reference_autograd(golden):
    g = golden('g10_gauss_net')
    for tag, eps in (('epsNone_', None), ('eps32_', 32.0)):
        gs = O.gauss_backward(g['s'], g['wi'], g['ori'], g['Gx'], g['Gr'], eps)
        assert rel_err(gs, g[tag + 'grad_s']) < 1e-5


def test_igsm_step_matches_reference(golden):
    g = golden('g11_igsm_step')
    for targeted in (False, True):
        out = O.igsm_step(g['s'], g['grad'], g['s_init'], 2.0, 32.0, targeted)
        assert np.array_equal(out, g['out_targeted%d' % int(targeted)])
