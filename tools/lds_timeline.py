#!/usr/bin/env python3
"""Reads the stamps of the `lds_timeline` experiment build (tools/experiment.py lds_timeline): where a SMALL launch of the
LDS-ring forward kernel (the training step's 2 and 6 tiles per wave) spends its time - launch skew across the grid, the
constant load, the ring start, every round's tile (wall clock AND shader cycles: their ratio is the clock the tile ran at),
the drain - against the HIP-event time of the same launch, under three launch cadences:
  sync   : one launch, then a host synchronize (gaps of host time between launches)
  queue  : 8 launches enqueued back to back, stamps of the last
  after  : a long launch of the same kernel (24 rounds) enqueued right in front of it

    python3 tools/lds_timeline.py [RxN ...]        (default 1024x64 1024x192 4096x192)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_lds_timeline.so')
import synth  # noqa: E402
from nerfail_amd import _train  # noqa: E402
from nerfail_amd.run_nerf import _mlp_points  # noqa: E402
from nerfail_amd.run_nerf_helpers import NeRF  # noqa: E402

dev = torch.device('cuda:0')
sd = synth.nerf_state_dict(seed=1)
m = NeRF(8, 256, 63, 27, 5, [4], True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev)
TICK_US = 0.01                                           # s_memrealtime: 100 MHz
big_pts = torch.randn((4096, 192, 3), device=dev)
big_vd = torch.nn.functional.normalize(torch.randn((4096, 3), device=dev), dim=-1)


def report(name, sz, R, N, raw, ev):
    ntiles = (R * N + 31) // 32
    waves = min(1024, (ntiles + 3) // 4 * 4)
    rounds = (ntiles + 1023) // 1024
    w = raw.reshape(-1).view(torch.int64)[:waves * 32].reshape(waves, 32).cpu().numpy()
    t0 = w[:, 0].min()
    us = lambda a: (a - t0) * TICK_US
    print('== %s %s: %d tiles, %d round(s); HIP events %.1f us' % (name, sz, ntiles, rounds, ev))
    print('   kernel entry: median %.1f, last %.1f us; constants +%.1f us, ring started +%.1f us' % (
        np.median(us(w[:, 0])), us(w[:, 0]).max(), np.median((w[:, 1] - w[:, 0]) * TICK_US), np.median((w[:, 2] - w[:, 1]) * TICK_US)))
    prev, prevc = w[:, 2], w[:, 16 + 2]
    for r in range(min(rounds, 12)):
        d = (w[:, 3 + r] - prev) * TICK_US
        c = (w[:, 16 + 3 + r] - prevc).astype(np.float64)
        print('   tile of round %-2d: median %.1f us (%.1f .. %.1f), %.0f shader cycles = %.0f MHz   (9280 MFMA x 64 = 593 920 cycles)' % (
            r, np.median(d), d.min(), d.max(), np.median(c), np.median(c / d)))
        prev, prevc = w[:, 3 + r], w[:, 16 + 3 + r]
    print('   kernel end: first %.1f, median %.1f, last %.1f us after the first entry' % (us(w[:, 15]).min(), np.median(us(w[:, 15])), us(w[:, 15]).max()))


for sz in (sys.argv[1:] or ['1024x64', '1024x192', '4096x192']):
    R, N = [int(v) for v in sz.split('x')]
    pts = torch.randn((R, N, 3), device=dev)
    vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
    for name, fn, big in (('inference', lambda: _mlp_points(m, pts, vd), lambda: _mlp_points(m, big_pts, big_vd)),
                          ('training forward', lambda: _train.mlp_fwd_train(m, pts, vd)[0], lambda: _train.mlp_fwd_train(m, big_pts, big_vd)[0])):
        for cadence in ('sync', 'queue', 'after'):
            for _ in range(3):
                raw = fn()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if cadence == 'queue':
                for _ in range(7):
                    fn()
            elif cadence == 'after':
                big()
            e0.record()
            raw = fn()
            e1.record()
            torch.cuda.synchronize()
            report('%s [%s]' % (name, cadence), sz, R, N, raw, e0.elapsed_time(e1) * 1e3)
