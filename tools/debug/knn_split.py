"""K8 on rendered-view geometry, split by query class: background pixels (near plane, far from the set) vs surface pixels."""
import sys, time, torch, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
from nerfail_amd.create_index_and_dist import index_and_dist
dev = torch.device("cuda:0")
H = W = 800
S = torch.from_numpy(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)).to(dev)
Q = torch.from_numpy(synth.sphere_view_points(H, W, 45.)).to(dev).reshape(-1, 3)
out = index_and_dist(Q.reshape(H, W, 3), S)
far = (out[0].reshape(-1, 8)[:, 0] > 0.5)
print('far share %.3f' % far.float().mean().item())
for name, sel in (('all', torch.ones_like(far)), ('background', far), ('surface', ~far)):
    q = Q[sel].contiguous()
    n = q.shape[0] // 64 * 64
    q = q[:n].reshape(-1, 64, 3).contiguous()
    index_and_dist(q, S); torch.cuda.synchronize(); t = time.time()
    for _ in range(3): index_and_dist(q, S)
    torch.cuda.synchronize(); print('%-11s %7d queries  %.2f ms' % (name, n, (time.time() - t) / 3 * 1e3))
q = Q[:64].reshape(1, 64, 3).contiguous()
index_and_dist(q, S); torch.cuda.synchronize(); t = time.time()
for _ in range(3): index_and_dist(q, S)
torch.cuda.synchronize(); print('grid build only (64 queries) %.2f ms' % ((time.time() - t) / 3 * 1e3))
