"""K8 on view geometry, split by query kind: surface hits (on the object) vs background pixels (near-plane points, far from
the set); kernel-only time via rocprof-free HIP events around the whole call (grid build included) and around a second
call with the same workspace. Usage: python tools/debug/knn_split.py"""
import sys, time, torch, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
from nerfail_amd.create_index_and_dist import index_and_dist
dev = torch.device("cuda:0")
H = W = 800
S = torch.from_numpy(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)).to(dev)
Qn = synth.sphere_view_points(H, W, 45.).reshape(-1, 3)
surf = np.linalg.norm(Qn, axis=1) < 1.25
print('queries: %d surface, %d background' % (surf.sum(), (~surf).sum()))


def t_ms(Q, reps=5):
    index_and_dist(Q, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        index_and_dist(Q, S)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


Qall = torch.from_numpy(Qn).to(dev).reshape(H, W, 3)
Qs = torch.from_numpy(np.ascontiguousarray(Qn[surf])).to(dev).reshape(1, -1, 3)
Qb = torch.from_numpy(np.ascontiguousarray(Qn[~surf])).to(dev).reshape(1, -1, 3)
Q8 = torch.from_numpy(np.ascontiguousarray(Qn[:8])).to(dev).reshape(1, -1, 3)
print('all %.2f ms   surface only %.2f ms   background only %.2f ms   8 queries (grid build) %.2f ms'
      % (t_ms(Qall), t_ms(Qs), t_ms(Qb), t_ms(Q8)))
S2 = torch.from_numpy(synth.sphere_shell_points(3 * H * W, seed=0)).to(dev)
Q2 = torch.from_numpy(synth.sphere_shell_points(H * W, seed=1).reshape(H, W, 3)).to(dev)
S_keep = S
S = S2
print('shell points %.2f ms' % t_ms(Q2))
