"""Time of one rendered-view map search (K8, 800 x 800 tiles) with the library named by NERFAIL_HIP_LIB (pricing builds of
tools/experiment.py knn_*: their results are wrong by construction). Usage: NERFAIL_HIP_LIB=... python tools/debug/knn_ablate.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import synth
from nerfail_amd.create_index_and_dist import index_and_dist
dev = torch.device('cuda:0')
H = W = 800
S = torch.from_numpy(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)).to(dev)
Q = torch.from_numpy(synth.sphere_view_points(H, W, 45.)).to(dev)
index_and_dist(Q, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    index_and_dist(Q, S)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print('%s: %.2f ms per view (min of 5; all %s)' % (os.path.basename(os.environ.get('NERFAIL_HIP_LIB', 'libnerfail_hip.so')), min(ts), ' '.join('%.2f' % t for t in ts)))
