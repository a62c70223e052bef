#!/usr/bin/env python3
"""DeepFool inner loop (deepfool.py:44-107 via nerfail_amd.deepfool) on one 800x800 view with the stand-in victim CNN:
time per iteration and where it goes (diagnostic)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import synth
import bench
from nerfail_amd.GaussNet import create_gauss_w, gauss_net
from nerfail_amd.deepfool import deepfool

dev = torch.device('cuda:0')
H = W = 800
P = 3
rs = np.random.RandomState(0)
Ns = P * H * W
base = (rs.randint(0, P, size=(1, 1, 1, 1)) * H * W + np.arange(H * W).reshape(1, H, W, 1))
idx = np.clip(base + rs.randint(-2 * W, 2 * W, size=(1, H, W, 8)), 0, Ns - 1).astype(np.float32)
dist_ = np.sort(np.abs(rs.normal(scale=0.02, size=(1, H, W, 8))).astype(np.float32), -1)
wi, _ = create_gauss_w(dev, 0.02)(torch.from_numpy(np.stack([dist_, idx], 1)).to(dev))
ori = torch.from_numpy(synth.disc_alpha_image(1, H, W, seed=3)).to(dev)
s = torch.zeros((P, H, W, 4), device=dev)
s[..., 3] = 255.0
torch.manual_seed(0)
victim = bench.victim_cnn(8).to(dev)
victim.requires_grad_(False)
net = gauss_net(dev, 0.02, victim, 'my_model', epsilon=None)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for rep in range(2):
    torch.cuda.synchronize()
    t = time.time()
    rot, loop_i, o, c, s_new = deepfool((s, wi, ori), 1.0, net, num_classes=8, max_iter=iters, target_label=None,
                                        overshoot=0.02, m1=1e6, m2=30)     # m1 huge: never breaks early
    torch.cuda.synchronize()
    dt = time.time() - t
    print('rep %d: %d iterations, %.2f ms per iteration (untargeted: 8 class gradients each)' % (rep, loop_i, dt / max(loop_i, 1) * 1e3))

# ---- segments of one iteration
def sync():
    torch.cuda.synchronize()
    return time.perf_counter()
st = s.clone().requires_grad_(True)
T_ = {}
for rep in range(4):
    t0 = sync()
    x, x_rgba, cla, _, ori_cla = net(st, wi, ori)
    t1 = sync()
    sel = torch.zeros((8, 1, 8), device=dev); sel[torch.arange(8), 0, torch.arange(8)] = 1.0
    try:
        J = torch.autograd.grad(cla, x_rgba, grad_outputs=sel, retain_graph=True, is_grads_batched=True)[0]
        mode = 'batched'
    except RuntimeError as e:
        J = torch.stack([torch.autograd.grad(cla, x_rgba, grad_outputs=sel[i], retain_graph=True)[0] for i in range(8)])
        mode = 'per-class (%s)' % str(e)[:80]
    t2 = sync()
    Jl = torch.stack([torch.autograd.grad(cla, x_rgba, grad_outputs=sel[i], retain_graph=True)[0] for i in range(8)])
    t3 = sync()
    G = net.logit_gradients(st, wi, x, x_rgba, cla, list(range(8)))
    t4 = sync()
    gp = G[1:] - G[0:1]
    nrm = torch.linalg.vector_norm(gp.reshape(7, -1), dim=1)
    t5 = sync()
    for k, v in (('forward(2 classifier fwd + K10)', t1 - t0), ('classifier bwd ' + mode, t2 - t1), ('classifier bwd per-class loop', t3 - t2),
                 ('logit_gradients total', t4 - t3), ('diff+norms', t5 - t4)):
        T_.setdefault(k, []).append(v * 1e3)
for k, v in T_.items():
    print('%-60s %.2f ms' % (k, float(np.median(v))))
