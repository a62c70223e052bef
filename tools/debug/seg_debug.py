import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from nerfail_amd.GaussNet import gauss_gather, csr_for
g = dict(np.load(os.path.join(ROOT, 'tests/golden/g10_gauss_net.npz')))
dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
out = {}
for det in (True, False):
    s = T(g['s']).requires_grad_(True)
    x, xr = gauss_gather(s, T(g['wi']), T(g['ori']), None, None, det)
    ((x * T(g['Gx'])).sum() + (xr * T(g['Gr'])).sum()).backward()
    out[det] = s.grad.reshape(-1, 4).cpu().numpy()
ref = g['epsNone_grad_s'].reshape(-1, 4)
bad = np.where(np.abs(out[True] - ref).max(1) > 1e-4 * np.abs(ref).max())[0]
print('bad rows', len(bad), bad[:20])
csr = csr_for(T(g['wi']), ref.shape[0])
rp = csr.row_ptr.cpu().numpy(); ro = csr.row_of.cpu().numpy()
print('E', rp[-1], 'n', len(ro))
for r in bad[:10]:
    print('row', r, 'entries', rp[r], rp[r + 1], 'len', rp[r + 1] - rp[r], 'chunk', rp[r] // 512, (rp[r + 1] - 1) // 512, 'pos in chunk', rp[r] % 512, 'got', out[True][r], 'ref', ref[r], 'atomic', out[False][r])
