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


def test_gauss_get_r_and_get_img_match_reference(golden):
    """g20: the reference's gauss_get_r.forward (GN:224-268) and the hot part of gauss_get_img.forward (GN:309-319)."""
    g = golden('g20_gauss_get')
    r, (emax, emin) = O.gauss_get_r(g['s'], g['dist_and_index'], c=0.02)
    assert rel_err(r, g['r']) < 1e-5
    assert (r[0, 0, 0] == 0).all()                            # sum g underflow -> 0 branch (GN:245)
    assert abs(emax - float(g['eps3d_max'])) < 1e-3 * abs(emax) and abs(emin - float(g['eps3d_min'])) < 1e-3 * abs(emin)
    x_rgba = O.gauss_get_img(g['ori'], g['r'])
    assert rel_err(x_rgba, g['x_rgba']) < 1e-6
    assert float(np.abs(x_rgba).max()) > 255.0 or float(x_rgba.min()) < 0.0     # the fixture exercises "no [0,255] clip"


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


def oracle_cfg3_loop(g):
    """The cfg3 loop (AS:278-392) with the numpy oracle for the pixel<->3-D map and torch-CPU for the stand-in
    classifier tail (GN:121-157) + CE; yields the perturbation rgb after every step. Shared with the GPU test."""
    import torch
    P, H, W, NB, B, ITERS = [int(v) for v in g['shape']]
    s0, ori, wi = g['s0'], g['ori'], g['wi']
    cls_w = torch.from_numpy(g['cls_w'])
    s = s0.copy()
    for it in range(ITERS):
        for b in range(NB):
            sl = slice(b * B, (b + 1) * B)
            x, x_rgba, _ = O.gauss_forward(s, wi[sl], ori[sl], None)
            xr = torch.from_numpy(x_rgba).requires_grad_(True)
            c = xr.permute(0, 3, 1, 2)
            img = torch.where(c[:, 3:4] > 0, c[:, :3], torch.full_like(c[:, :3], 255.))
            cla = torch.nn.functional.adaptive_avg_pool2d(img, 4).reshape(B, -1) @ cls_w.t()
            loss = torch.nn.functional.cross_entropy(cla, torch.full((B,), int(g['label']), dtype=torch.long))
            loss.backward()
            grad = O.gauss_backward(s, wi[sl], ori[sl], np.zeros_like(x), xr.grad.numpy(), None)
            s = O.igsm_step(s, grad, s0, float(g['a']), float(g['epsilon']), False)
            yield it, b, s, float(loss.detach())


def test_cfg3_loop_matches_reference_iterates(golden):
    """Fixture g15: 20 iterations x 2 batches of 8 views through the reference's gauss_net + the re-issued sign step.
    The update is a sign step: an element may differ from the reference only where its gradient is at rounding level,
    and such a flip moves later iterates by one step of a = 2 at that element."""
    g = golden('g15_cfg3_loop')
    ref = g['iterates_rgb_int8'].astype(np.float32)
    worst, step = 0.0, 0
    for it, b, s, loss in oracle_cfg3_loop(g):
        diff = s[..., :3] != ref[step]
        worst = max(worst, float(diff.mean()))
        assert np.abs(s[..., :3] - ref[step]).max() <= 2 * float(g['a']), (it, b)
        assert abs(loss - g['losses'][step]) <= 2e-4 * abs(g['losses'][step]), (it, b, loss, g['losses'][step])
        step += 1
    assert step == 40
    print('oracle vs reference cfg3 iterates: worst fraction of differing elements %.2e' % worst)
    assert worst < 2e-3
