#!/usr/bin/env python3
"""Kernel-level A/B timings of the NeRF MLP kernels at training-step sizes (HIP events, interleaved rounds)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import synth  # noqa: E402
from nerfail_amd import _lib, _train  # noqa: E402
from nerfail_amd.run_nerf import _mlp_points  # noqa: E402
from nerfail_amd.run_nerf_helpers import NeRF  # noqa: E402

dev = torch.device('cuda:0')
FLOP = 2 * 593408


def net(seed):
    sd = synth.nerf_state_dict(seed=seed)
    m = NeRF(8, 256, 63, 27, 5, [4], True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev)


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=None, help='alternative libnerfail_hip build (tools/experiment.py)')
    ap.add_argument('--only', default=None, help='comma-separated kernel names')
    ap.add_argument('--sizes', default='1024x64,1024x192,8192x192')
    ap.add_argument('--dual', default=None, help='RxNc+Nf: time the two-network backward launches, e.g. 1024x64+192')
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    only = set(args.only.split(',')) if args.only else None
    m = net(1)
    if args.dual:                     # the training step's backward: coarse + fine network in ONE launch per kernel
        R, rest = args.dual.split('x')
        R, (Nc, Nf) = int(R), [int(v) for v in rest.split('+')]
        m1 = net(2)
        M0, M1 = R * Nc, R * Nf
        pts = torch.randn((R, Nc + Nf, 3), device=dev)
        vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
        nA = _train.acts_floats(m, M0)
        acts = torch.empty((nA + _train.acts_floats(m1, M1),), device=dev)
        _train.mlp_fwd_train(m, pts[:, :Nc].contiguous(), vd, acts=acts[:nA])
        _train.mlp_fwd_train(m1, pts[:, Nc:].contiguous(), vd, acts=acts[nA:])
        d_raw = torch.randn((M0 + M1, 4), device=dev) * 1e-3
        lib = _lib.load()
        dz = torch.empty((_train.dz_floats(m, M0) + _train.dz_floats(m1, M1),), device=dev)
        (p0, pT0), (p1, pT1) = _train.packed_both(m), _train.packed_both(m1)
        g0, g1 = _train._new_grads(m, False), _train._new_grads(m1, False)
        sc, nb = _train.dw_scratch(m, M0, M1, 0, dev)

        def f_bd2():
            _lib.check(lib.nerfail_mlp_bwd_data2(_lib.dev(p0), _lib.dev(pT0), M0, _lib.dev(p1), _lib.dev(pT1), M1, m.D, m.W, m._skip(),
                                                 _lib.dev(d_raw), _lib.dev(acts), _lib.dev(dz), _lib.stream()))

        def f_bw2():
            _lib.check(lib.nerfail_mlp_bwd_weights(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), M0, _train._grads_struct(m, g0), M1,
                                                   _train._grads_struct(m1, g1), 0, _lib.dev(sc), nb, _lib.stream()))
        for name, fn in (('bwd_data2', f_bd2), ('bwd_weights2', f_bw2)):
            med, mn = timeit(fn)
            print('M=%d+%d %-12s median %8.3f ms  min %8.3f ms  -> %6.1f TFLOP/s (fwd-equivalent FLOPs)' %
                  (M0, M1, name, med, mn, (M0 + M1) * FLOP / (med * 1e-3) / 1e12), flush=True)
        return
    for R, N in [tuple(int(v) for v in sz.split('x')) for sz in args.sizes.split(',')]:
        M = R * N
        pts = torch.randn((R, N, 3), device=dev)
        vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
        d_raw = torch.randn((R, N, 4), device=dev) * 1e-3
        raw, acts = _train.mlp_fwd_train(m, pts, vd)
        grads = [torch.zeros_like(p) for p in _train.ordered_params(m)]
        lib = _lib.load()
        dz = torch.empty((lib.nerfail_mlp_train_dz_floats(m.D, m.W, M),), device=dev)
        pk, pkT = m.packed(), _train.packed_T(m)

        def f_inf():
            _mlp_points(m, pts, vd)

        m16 = net(1)
        m16.precision = 'f16x3'

        def f_f16():
            _mlp_points(m16, pts, vd)

        def f_train():
            _train.mlp_fwd_train(m, pts, vd)

        def f_bd():
            _lib.check(lib.nerfail_mlp_bwd_data(_lib.dev(pk), _lib.dev(pkT), m.D, m.W, m._skip(), _lib.dev(d_raw),
                                                _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))

        sc0, nb0 = _train.dw_scratch(m, M, 0, 0, dev)
        sc1, nb1 = _train.dw_scratch(m, M, 0, _lib.DW_BF16X3, dev)

        def f_bw():
            _lib.check(lib.nerfail_mlp_bwd_weights(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), M,
                                                   _train._grads_struct(m, grads), 0, None, 0, _lib.dev(sc0), nb0, _lib.stream()))
        pk16T = _train.packed_f16_T(m16)

        def f_train16():
            _train.mlp_fwd_train(m16, pts, vd)

        def f_bd16():
            _lib.check(lib.nerfail_mlp_bwd_data_f16(_lib.dev(pk), _lib.dev(pk16T), m.D, m.W, m._skip(), _lib.dev(d_raw),
                                                    _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))

        def f_bw16():
            _lib.check(lib.nerfail_mlp_bwd_weights(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), M,
                                                   _train._grads_struct(m, grads), 0, None, _lib.DW_BF16X3, _lib.dev(sc1), nb1, _lib.stream()))
        for name, fn in (('fwd_infer', f_inf), ('fwd_f16x3', f_f16), ('fwd_train', f_train), ('fwd_train_f16', f_train16),
                         ('bwd_data_f16', f_bd16), ('bwd_w_bf16x3', f_bw16), ('bwd_data', f_bd), ('bwd_weights', f_bw)):
            if only and name not in only:
                continue
            med, mn = timeit(fn)
            print('M=%7d %-12s median %8.3f ms  min %8.3f ms  -> %6.1f TFLOP/s (fwd-equivalent FLOPs)' %
                  (M, name, med, mn, M * FLOP / (med * 1e-3) / 1e12), flush=True)


if __name__ == '__main__':
    main()
