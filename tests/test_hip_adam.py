"""K13 fused Adam through the C ABI vs the reference's optimizer (fixture g13) and vs the oracle."""
import numpy as np
import pytest
import torch

from hiputil import T, N
from oracle import adam as A

pytestmark = pytest.mark.gpu


def _run(g, make_opt):
    n, steps = int(g['n_tensors']), int(g['n_steps'])
    params = [torch.nn.Parameter(T(g['init_%d' % i].copy())) for i in range(n)]
    opt = make_opt(params)
    lrate, decay = float(g['lrate']), int(g['lrate_decay'])
    gs, hist = 0, []
    for it in range(steps):
        for i, p in enumerate(params):
            p.grad = T(g['g%d_%d' % (it, i)].copy())
        assert float(opt.param_groups[0]['lr']) == float(g['lr_%d' % it])
        opt.step()
        for param_group in opt.param_groups:                       # the reference's lines RN:796-800, unchanged
            param_group['lr'] = A.decayed_lrate(lrate, gs, decay)
        gs += 100000 if it == 2 else 1
        hist.append([(N(p).copy(), N(opt.state[p]['exp_avg']).copy(), N(opt.state[p]['exp_avg_sq']).copy()) for p in params])
    return hist, opt, params


def test_adam_matches_reference(golden):
    from nerfail_amd.optim import Adam
    g = golden('g13_adam')
    hist, opt, params = _run(g, lambda ps: Adam(params=ps, lr=float(g['lrate']), betas=(0.9, 0.999)))
    total = differing = 0
    for it, per in enumerate(hist):
        for i, (p, m, v) in enumerate(per):
            assert np.array_equal(m, g['m%d_%d' % (it, i)]), (it, i)
            assert np.array_equal(v, g['v%d_%d' % (it, i)]), (it, i)
            ref = g['p%d_%d' % (it, i)]
            assert np.all(np.abs(p - ref) <= np.spacing(np.abs(ref).astype(np.float32))), (it, i)
            total += p.size
            differing += int((p != ref).sum())
    assert differing <= total // 500, (differing, total)
    # state layout is torch's: a reference checkpoint's optimizer_state_dict loads (RN:219) and round-trips
    sd = opt.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == len(hist)
    stock = torch.optim.Adam(params, lr=1e-3)
    stock.load_state_dict(sd)
    opt2 = Adam(params=params, lr=1e-3)
    opt2.load_state_dict(stock.state_dict())
    assert float(opt2.state[params[0]]['step']) == len(hist)


def test_adam_many_tensors_and_argument_checks():
    """More tensors than one launch's table holds (48), odd sizes, an empty tensor; plus the ABI's error paths."""
    from nerfail_amd import _lib
    from nerfail_amd.optim import Adam
    rng = np.random.default_rng(5)
    shapes = [(int(rng.integers(1, 700)),) for _ in range(60)] + [(0,), (300, 7)]
    p0 = [rng.normal(size=s).astype(np.float32) for s in shapes]
    g0 = [rng.normal(size=s).astype(np.float32) * 1e-2 for s in shapes]
    params = [torch.nn.Parameter(T(a.copy())) for a in p0]
    opt = Adam(params=params, lr=5e-4, betas=(0.9, 0.999))
    for k in range(2):
        for p, g in zip(params, g0):
            p.grad = T(g * (k + 1))
        opt.step()
    for a, g, p in zip(p0, g0, params):
        m, v = np.zeros_like(a), np.zeros_like(a)
        for k in range(2):
            a, m, v = A.adam_step(a, g * np.float32(k + 1), m, v, 5e-4, k + 1)
        assert np.all(np.abs(N(p) - a) <= np.spacing(np.abs(a))), p.shape
    lib = _lib.load()
    assert lib.nerfail_adam_step(None, 0, 0.9, 0.999, 1e-8, None) == 0
    assert lib.nerfail_adam_step(None, 3, 0.9, 0.999, 1e-8, None) != 0
    e = _lib.AdamTensor()
    e.numel = 4                                                     # NULL pointers with numel > 0
    e.bias_correction2_sqrt = 1.0
    assert lib.nerfail_adam_step((_lib.AdamTensor * 1)(e), 1, 0.9, 0.999, 1e-8, None) != 0
    assert b'NULL' in lib.nerfail_last_error()
    with pytest.raises(NotImplementedError):
        Adam(params=params, lr=1e-3, weight_decay=0.1).step()
