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
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    only = set(args.only.split(',')) if args.only else None
    m = net(1)
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

        def f_bw():
            _lib.check(lib.nerfail_mlp_bwd_weights(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), M,
                                                   _train._grads_struct(m, grads), _lib.stream()))
        pk16T = _train.packed_f16_T(m16)

        def f_train16():
            _train.mlp_fwd_train(m16, pts, vd)

        def f_bd16():
            _lib.check(lib.nerfail_mlp_bwd_data_f16(_lib.dev(pk), _lib.dev(pk16T), m.D, m.W, m._skip(), _lib.dev(d_raw),
                                                    _lib.dev(acts), M, _lib.dev(dz), _lib.stream()))

        def f_bw16():
            _lib.check(lib.nerfail_mlp_bwd_weights_bf16x3(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), M,
                                                          _train._grads_struct(m, grads), _lib.stream()))
        for name, fn in (('fwd_infer', f_inf), ('fwd_f16x3', f_f16), ('fwd_train', f_train), ('fwd_train_f16', f_train16),
                         ('bwd_data_f16', f_bd16), ('bwd_w_bf16x3', f_bw16), ('bwd_data', f_bd), ('bwd_weights', f_bw)):
            if only and name not in only:
                continue
            med, mn = timeit(fn)
            print('M=%7d %-12s median %8.3f ms  min %8.3f ms  -> %6.1f TFLOP/s (fwd-equivalent FLOPs)' %
                  (M, name, med, mn, M * FLOP / (med * 1e-3) / 1e12), flush=True)


if __name__ == '__main__':
    main()
