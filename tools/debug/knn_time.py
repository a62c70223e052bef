import sys, time, torch, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,"tests")]
import synth
from nerfail_amd.create_index_and_dist import index_and_dist
dev=torch.device("cuda:0")
H=W=800
S=torch.from_numpy(np.stack([synth.sphere_view_points(H,W,th) for th in (-120.,0.,120.)]).reshape(-1,3)).to(dev)
for th in (-171., 33.):
    Q=torch.from_numpy(synth.sphere_view_points(H,W,th)).to(dev)
    out=index_and_dist(Q,S); torch.cuda.synchronize(); t=time.time()
    for _ in range(3): out=index_and_dist(Q,S)
    torch.cuda.synchronize(); print("sphere view", th, "ms per view %.2f" % ((time.time()-t)/3*1e3))
    rows=[0, 137, 300, 400, 650, 799]
    ob=torch.stack([index_and_dist(Q[r:r+1],S,method="brute")[:,0] for r in rows],1)
    print("  rows", rows, "equal to brute force:", bool(torch.equal(out[:,rows], ob)))
S2=torch.from_numpy(synth.sphere_shell_points(3*H*W,seed=0)).to(dev); Q2=torch.from_numpy(synth.sphere_shell_points(H*W,seed=1).reshape(H,W,3)).to(dev)
out=index_and_dist(Q2,S2); torch.cuda.synchronize(); t=time.time()
for _ in range(3): out=index_and_dist(Q2,S2)
torch.cuda.synchronize(); print("shell points ms per view %.2f" % ((time.time()-t)/3*1e3))
