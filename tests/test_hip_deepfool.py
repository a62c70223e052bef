"""-m gpu: deepfool (deepfool.py:10-111, the NeRFail inner loop) over the HIP gauss path vs the reference's own run
(fixture g12: same inputs, same stand-in classifier weights)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from hiputil import T, N, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag,target,over,iters', [('untargeted', None, 0.02, 6), ('targeted', 2, 0.02, 6),
                                                   ('untargeted_break', None, 1.0, 12), ('targeted_break', 2, 1.0, 12)])
def test_deepfool_matches_reference(golden, tag, target, over, iters):
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.deepfool import deepfool
    g = golden('g12_deepfool')
    w = T(g['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ w.t()
    net = gauss_net(dev(), 0.02, Cls(), 'my_model', epsilon=None)
    rot, loop_i, ori_idx, cla_idx, s_new = deepfool((T(g['s']), T(g['wi']), T(g['ori'])), 1.0, net, num_classes=8,
                                                    max_iter=iters, target_label=target, overshoot=over, m1=0.05, m2=0.5)
    assert loop_i == int(g[tag + '_loop_i'])
    assert int(ori_idx) == int(g[tag + '_ori_idx']) and int(cla_idx) == int(g[tag + '_cla_idx'])
    assert rel_err(N(rot), g[tag + '_rot']) < 1e-3            # 6-12 chained gradient steps in fp32
    assert rel_err(N(s_new), g[tag + '_s_new']) < 1e-3
    assert np.array_equal(N(s_new)[..., 3], g['s'][..., 3])   # alpha channel untouched
