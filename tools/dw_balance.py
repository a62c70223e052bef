#!/usr/bin/env python3
"""Per-workgroup busy time of the LDS-staged weight-gradient kernel (needs `python tools/experiment.py dw_clock`:
every workgroup writes its wall-clock ticks over the start of dz) in the training-step configuration (coarse 1024x64 +
fine 1024x192 samples in ONE launch) and a least-squares fit of ns per TILE for each group shape plus a constant per
segment: the source of kTileNs in nerfail_amd/csrc/mlp_dw.hip. Usage: python tools/dw_balance.py [RxNc+Nf]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['NERFAIL_HIP_LIB'] = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_dw_clock.so')
os.environ['NERFAIL_DW_KERNEL'] = 'lds'
import synth
from nerfail_amd import _lib, _train
from nerfail_amd.run_nerf_helpers import NeRF
dev = torch.device('cuda:0')
K_TILE_NS = [[7000, 3800, 2200, 800, 780, 1200], [2643, 1717, 1361, 933, 924, 1115]]   # copy of kTileNs (mlp_dw.hip)
R, NC, NF = 1024, 64, 192


def net(seed):
    sd = synth.nerf_state_dict(seed=seed)
    m = NeRF(8, 256, 63, 27, 5, [4], True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev)


def groups_of_one_net():
    """(shape) of every group in descriptor order (build_descs in mlp_dw.hip), D = 8, skip = 4."""
    table = {(8, 8): 0, (4, 8): 1, (8, 2): 2, (4, 1): 3, (1, 4): 4, (1, 8): 5}
    g = []
    for i in range(8):
        if i == 0 or i == 5:
            g.append(table[(8, 2)])
        if i > 0:
            g.append(table[(8, 8)])
    g += [table[(8, 8)], table[(4, 8)], table[(4, 1)], table[(1, 4)], table[(1, 8)]]
    return g


def partition(bf16, tiles, wgs=256):
    """A[wg, shape] = tiles of that shape the workgroup processes, A[wg, 6] = its number of segments."""
    shapes, ntile = [], []
    for t in tiles:
        if t:
            shapes += groups_of_one_net()
            ntile += [t] * len(groups_of_one_net())
    cost = [K_TILE_NS[1 if bf16 else 0][sh] for sh in shapes]
    cum = np.concatenate([[0], np.cumsum([c * t for c, t in zip(cost, ntile)])]).astype(np.int64)
    total = int(cum[-1])
    A = np.zeros((wgs, 7))
    for b in range(wgs):
        lo = total // wgs * b + (total % wgs) * b // wgs
        hi = total // wgs * (b + 1) + (total % wgs) * (b + 1) // wgs
        for g, sh in enumerate(shapes):
            g0, g1 = int(cum[g]), int(cum[g + 1])
            if hi <= g0 or lo >= g1:
                continue
            c = cost[g]
            s_, e_ = max(lo, g0) - g0, min(hi, g1) - g0
            tb, te = -(-s_ // c), -(-e_ // c)
            if te > tb:
                A[b, sh] += te - tb
                A[b, 6] += 1
    return A


def main():
    m0, m1 = net(1), net(2)
    M0, M1 = R * NC, R * NF
    pts = torch.randn((R, NC + NF, 3), device=dev)
    vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
    acts = torch.empty((_train.acts_floats(m0, M0) + _train.acts_floats(m1, M1),), device=dev)
    n0 = _train.acts_floats(m0, M0)
    _train.mlp_fwd_train(m0, pts[:, :NC].contiguous(), vd, acts=acts[:n0])
    _train.mlp_fwd_train(m1, pts[:, NC:].contiguous(), vd, acts=acts[n0:])
    d_raw = torch.randn((M0 + M1, 4), device=dev) * 1e-3
    lib = _lib.load()
    for bf16 in (0, 1):
        flags = _lib.DW_BF16X3 if bf16 else 0
        ts = []
        for rep in range(5):
            dz = torch.empty((_train.dz_floats(m0, M0) + _train.dz_floats(m1, M1),), device=dev)
            p0, pT0 = _train.packed_both(m0)
            p1, pT1 = _train.packed_both(m1)
            _lib.check(lib.nerfail_mlp_bwd_data2(_lib.dev(p0), _lib.dev(pT0), M0, _lib.dev(p1), _lib.dev(pT1), M1, 8, 256, 4,
                                                 _lib.dev(d_raw), _lib.dev(acts), _lib.dev(dz), _lib.stream()))
            g0, g1 = _train._new_grads(m0, False), _train._new_grads(m1, False)
            scratch, nb = _train.dw_scratch(m0, M0, M1, flags, dev)
            _lib.check(lib.nerfail_mlp_bwd_weights(8, 256, 4, _lib.dev(acts), _lib.dev(dz), M0, _train._grads_struct(m0, g0), M1,
                                                   _train._grads_struct(m1, g1), flags, _lib.dev(scratch), nb, _lib.stream()))
            torch.cuda.synchronize()
            ts.append(dz.view(torch.int64)[:256].cpu().numpy().astype(np.float64) * 10.0)      # 100 MHz ticks -> ns
        t = np.median(np.stack(ts), 0)
        order = np.argsort(t)
        print('%s: per-workgroup busy us  min %.1f  mean %.1f  max %.1f  (max/mean %.3f)  slowest WGs %s  fastest %s'
              % ('bf16x3' if bf16 else 'f32', t.min() / 1e3, t.mean() / 1e3, t.max() / 1e3, t.max() / t.mean(), order[-4:], order[:4]))
        A = partition(bf16, (M0 // 32, M1 // 32))
        x, res, rank, sv = np.linalg.lstsq(A, t, rcond=None)
        print('   fitted ns per tile by shape %s, per segment %.0f ns; residual rms %.1f us'
              % (np.round(x[:6]).astype(int).tolist(), x[6], float(np.sqrt(np.mean((A @ x - t) ** 2))) / 1e3))
        print('   kTileNs row to use (segment constant folded in is NOT possible; keep it in mind): %s' % np.round(x[:6]).astype(int).tolist())


if __name__ == '__main__':
    main()
