"""Oracle (oracle/adam.py) vs the reference's optimizer run on CPU (fixture g13, tests/golden/make_golden.py)."""
import numpy as np

from oracle import adam as A


def test_adam_oracle_matches_reference(golden):
    g = golden('g13_adam')
    n, steps = int(g['n_tensors']), int(g['n_steps'])
    total = differing = 0
    for i in range(n):
        p = g['init_%d' % i].copy()
        m, v = np.zeros_like(p), np.zeros_like(p)
        for it in range(steps):
            p, m, v = A.adam_step(p, g['g%d_%d' % (it, i)], m, v, float(g['lr_%d' % it]), it + 1)
            assert np.array_equal(m, g['m%d_%d' % (it, i)])
            assert np.array_equal(v, g['v%d_%d' % (it, i)])
            ref = g['p%d_%d' % (it, i)]
            ulp = np.spacing(np.abs(ref).astype(np.float32))
            assert np.all(np.abs(p - ref) <= ulp), (i, it)
            total += p.size
            differing += int((p != ref).sum())
    assert differing <= total // 500, (differing, total)           # bit-exact but for rare 1-ulp cases


def test_lr_decay_matches_reference(golden):
    g = golden('g13_adam')
    lrate, decay = float(g['lrate']), int(g['lrate_decay'])
    gs, lrs = 0, [lrate]
    for it in range(int(g['n_steps']) - 1):
        lrs.append(A.decayed_lrate(lrate, gs, decay))
        gs += 100000 if it == 2 else 1
    assert lrs == [float(g['lr_%d' % it]) for it in range(int(g['n_steps']))]
