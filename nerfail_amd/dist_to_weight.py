"""MI355X mirror of tools/dist_to_weight.py (reference = DW; the script runs at import there, here it is a function).

Reads `index_and_dist/<split>/<i>.pth` (float32 [2,H,W,8]: distances, indices-as-float; written by
create_index_and_dist, CI:148-163), applies create_gauss_w (GN:169-186, K9 in libnerfail_hip.so) and writes
`index_and_weight/<split>/<i>.pth` (float32 [2,H,W,8]: weights, indices) - DW:82-97. Returns the mean squared distance
the reference prints as "v" (DW:92-93, :99-100). Files are saved as CPU tensors (the reference pickles them on
whatever device it ran on and loads with map_location, MyDataset.py:201; CPU loads everywhere)."""
import os

import torch

from .GaussNet import create_gauss_w


def dist_to_weight(label, basedir='../Create_spatial_point_set', test_number=200, val_number=100, train_number=100, c=0.02,
                   device=None):
    root = os.path.join(basedir, 'logs', 'blender_paper_' + label)
    src = os.path.join(root, 'index_and_dist')
    dst = os.path.join(root, 'index_and_weight')
    net = create_gauss_w(device, c)
    v_list = []
    for split, n in (('test', test_number), ('val', val_number), ('train', train_number)):       # DW:66-73 order
        os.makedirs(os.path.join(dst, split), exist_ok=True)
        for i in range(n):
            dai = torch.load(os.path.join(src, split, '%d.pth' % i), map_location='cpu')
            i_w, dist = net(dai.unsqueeze(0))                                                   # batch_size=1 (DW:78)
            v_list.append(float(torch.mean(torch.square(dist))))
            torch.save(i_w.squeeze(0).cpu(), os.path.join(dst, split, '%d.pth' % i))
    v = sum(v_list) / len(v_list)
    print('v :', v)
    return v
