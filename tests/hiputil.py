"""Helpers shared by the -m gpu parity tests (HIP path through the C ABI vs oracle / golden)."""
import numpy as np
import torch

import synth


def dev():
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


def hip_nerf(D=8, W=256, seed=0, requires_grad=False, precision='f32'):
    from nerfail_amd.run_nerf_helpers import NeRF
    sd = synth.nerf_state_dict(D=D, W=W, seed=seed)
    m = NeRF(D=D, W=W, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.requires_grad_(requires_grad)       # forward tests: inference path; training tests ask for gradients
    m.precision = precision
    return sd, m.to(dev())
