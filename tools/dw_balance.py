#!/usr/bin/env python3
"""Per-workgroup busy time of the LDS-staged weight-gradient kernel (needs `python tools/experiment.py dw_clock`:
every workgroup writes its wall-clock ticks over the start of dz). Shows how well the cost model balances the grid."""
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
sd = synth.nerf_state_dict(seed=1)
m = NeRF(8, 256, 63, 27, 5, [4], True)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.to(dev)
R, N = 1024, 192
pts = torch.randn((R, N, 3), device=dev)
vd = torch.nn.functional.normalize(torch.randn((R, 3), device=dev), dim=-1)
d_raw = torch.randn((R, N, 4), device=dev) * 1e-3
raw, acts = _train.mlp_fwd_train(m, pts, vd)
lib = _lib.load()
for bf16 in (0, 1):
    dz = torch.empty((lib.nerfail_mlp_train_dz_floats(m.D, m.W, R * N),), device=dev)
    _lib.check(lib.nerfail_mlp_bwd_data(_lib.dev(m.packed()), _lib.dev(_train.packed_T(m)), m.D, m.W, m._skip(), _lib.dev(d_raw),
                                        _lib.dev(acts), R * N, _lib.dev(dz), _lib.stream()))
    grads = _train._zero_grads(m)
    fn = lib.nerfail_mlp_bwd_weights_bf16x3 if bf16 else lib.nerfail_mlp_bwd_weights
    _lib.check(fn(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), R * N, _train._grads_struct(m, grads), _lib.stream()))
    torch.cuda.synchronize()
    t = dz.view(torch.int64)[:256].cpu().numpy().astype(np.float64) * 1e-5      # ms
    order = np.argsort(t)
    print('%s: per-workgroup busy ms  min %.3f  mean %.3f  max %.3f   slowest WGs %s  fastest %s' %
          ('bf16x3' if bf16 else 'f32', t.min(), t.mean(), t.max(), order[-4:], order[:4]))

# ---- fit the per-step cost of each group shape from the per-workgroup times (least squares)
K_STEP_NS = [[4460, 2450, 1486, 496, 483, 732], [2030, 1427, 987, 551, 539, 648]]   # copy of kStepNs (mlp_bwd.hip)
def partition(bf16):
    NT, D = 8, 8
    groups = []      # (shape, ma*nb, G)
    def add(LA, LB):
        table = {(8, 8): (0, 4, 4), (4, 8): (1, 4, 2), (8, 2): (2, 2, 2), (4, 1): (3, 1, 1), (1, 4): (4, 1, 1), (1, 8): (5, 1, 2)}
        sh, ma, nb = table[(LA, LB)]
        groups.append((sh, ma * nb, (2 * (LA + LB) + 3) // 4))
    for i in range(D):
        emb = (i == 0) or (i == 5)
        if emb:
            add(8, 2)
        if i > 0:
            add(8, 8)
    add(8, 8)            # feature
    add(4, 8); add(4, 1)  # views: feature part, dir part
    add(1, 4)            # rgb
    add(1, 8)            # alpha
    ntiles = R * N // 32
    cost = [K_STEP_NS[1 if bf16 else 0][sh] for sh, _, _ in groups]          # must mirror kStepNs in mlp_bwd.hip
    cum = np.concatenate([[0], np.cumsum([c * ntiles for c in cost])])
    total, W = cum[-1], 256
    A = np.zeros((W, 7))
    for b in range(W):
        lo = total // W * b + (total % W) * b // W
        hi = total // W * (b + 1) + (total % W) * (b + 1) // W
        for g, (sh, mn, G) in enumerate(groups):
            g0, g1 = cum[g], cum[g + 1]
            if hi <= g0 or lo >= g1:
                continue
            c = cost[g]
            s_, e_ = max(lo, g0) - g0, min(hi, g1) - g0
            tb, te = -(-s_ // c), -(-e_ // c)
            if te > tb:
                A[b, sh] += 2 * (te - tb)       # k16-steps
                A[b, 6] += 1                    # one accumulator flush per segment
    return A

for bf16 in (0, 1):
    dz = torch.empty((lib.nerfail_mlp_train_dz_floats(m.D, m.W, R * N),), device=dev)
    _lib.check(lib.nerfail_mlp_bwd_data(_lib.dev(m.packed()), _lib.dev(_train.packed_T(m)), m.D, m.W, m._skip(), _lib.dev(d_raw),
                                        _lib.dev(acts), R * N, _lib.dev(dz), _lib.stream()))
    ts = []
    for rep in range(3):
        grads = _train._zero_grads(m)
        fn = lib.nerfail_mlp_bwd_weights_bf16x3 if bf16 else lib.nerfail_mlp_bwd_weights
        _lib.check(fn(m.D, m.W, m._skip(), _lib.dev(acts), _lib.dev(dz), R * N, _train._grads_struct(m, grads), _lib.stream()))
        torch.cuda.synchronize()
        ts.append(dz.view(torch.int64)[:256].cpu().numpy().astype(np.float64) * 10.0)      # ns
    t = np.median(np.stack(ts), 0)
    A = partition(bf16)
    x, *_ = np.linalg.lstsq(A, t, rcond=None)
    print('bf16x3' if bf16 else 'f32', 'fitted ns per k16-step by shape [full, views, emb, dir, rgb, alpha], per flush:', np.round(x, 1),
          ' table now:', K_STEP_NS[1 if bf16 else 0])
