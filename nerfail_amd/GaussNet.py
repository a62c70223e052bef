"""MI355X mirror of model/GaussNet.py (reference = GN): the differentiable pixel <-> 3-D-point map.

gauss_net.forward keeps its signature and 5-tuple return. The hot part (GN:53-119: 8-NN gather, weighted
sum, alpha, epsilon clip, composite onto the image) is one HIP kernel with a hand-written backward
(scatter-add with float atomics); the cold tail (GN:121-157: layout change, white background, Resize,
classifier) stays stock PyTorch, as SURVEY.md section 8(a16) scopes it.
"""
import torch
from torch import nn

from . import _lib
from .run_nerf_helpers import _cuda


class _GaussGather(torch.autograd.Function):
    """x, x_rgba = f(spatial_rgb); d/d(spatial_rgb) by nerfail_gauss_bwd. weight/index/ori carry no grad."""

    @staticmethod
    def forward(ctx, spatial, wi, ori, epsilon, eps_minmax):
        dev = spatial.device
        s = _lib.f32c(spatial).reshape(-1, 4)
        B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
        x = torch.empty(tuple(ori.shape), dtype=torch.float32, device=dev)
        x_rgba = torch.empty(tuple(ori.shape), dtype=torch.float32, device=dev)
        eps = -1.0 if epsilon is None else float(epsilon)
        _lib.check(_lib.load().nerfail_gauss_fwd(_lib.dev(s, 'spatial_rgb'), s.shape[0], _lib.dev(wi, 'weight_and_index'),
                                                 _lib.dev(ori, 'ori_img'), B, P, eps, _lib.dev(x), _lib.dev(x_rgba),
                                                 _lib.dev(eps_minmax), _lib.stream()))
        ctx.save_for_backward(wi, ori, x)
        ctx.eps = eps
        ctx.s_shape = tuple(spatial.shape)
        return x, x_rgba

    @staticmethod
    def backward(ctx, grad_x, grad_x_rgba):
        wi, ori, x = ctx.saved_tensors
        B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
        n = 1
        for d in ctx.s_shape[:-1]:
            n *= d
        gs = torch.zeros((n, 4), dtype=torch.float32, device=x.device)
        gx = _lib.f32c(grad_x) if grad_x is not None else None
        gr = _lib.f32c(grad_x_rgba) if grad_x_rgba is not None else None
        _lib.check(_lib.load().nerfail_gauss_bwd(_lib.dev(wi), _lib.dev(ori), _lib.dev(x), _lib.dev(gx), _lib.dev(gr),
                                                 n, B, P, ctx.eps, _lib.dev(gs), _lib.stream()))
        return gs.reshape(ctx.s_shape), None, None, None, None


def gauss_gather(spatial_rgb, weight_and_index_list, ori_img, epsilon=None, eps_minmax=None):
    """Functional form of the hot part: returns (x, x_rgba), differentiable w.r.t. spatial_rgb."""
    dev = _cuda()
    wi = _lib.f32c(weight_and_index_list, dev)
    ori = _lib.f32c(ori_img, dev)
    if wi.dim() != 5 or wi.shape[1] != 2 or wi.shape[4] != 8:
        raise ValueError('weight_and_index_list must be [B,2,H,W,8] (DW:95-97)')
    if spatial_rgb.device != dev:
        spatial_rgb = spatial_rgb.to(dev)
    return _GaussGather.apply(spatial_rgb, wi, ori, epsilon, eps_minmax)


class gauss_net(nn.Module):
    """GN:8-159."""

    def __init__(self, device, c, model, model_name, epsilon=None):
        super(gauss_net, self).__init__()
        self.top_number = 8
        self.c = torch.nn.Parameter(torch.tensor([c]), requires_grad=False)
        self.device = device
        self.model = model
        self.model_name = model_name
        self.epsilon = epsilon
        self.update_epsilon_3d = True
        self._eps_minmax = None      # device-side running [min, max] of x_rgb*alpha (GN:89-103), read lazily

    # -- resize of the cold tail: torchvision if present (as the reference), else the same bilinear op in torch
    def _resize(self, x, size):
        try:
            from torchvision.transforms import Resize
            return Resize([size, size])(x)
        except ImportError:
            return torch.nn.functional.interpolate(x, size=(size, size), mode='bilinear', align_corners=False,
                                                   antialias=True)

    def _mm(self):
        if self._eps_minmax is None:
            self._eps_minmax = torch.zeros(2, dtype=torch.float32, device=_cuda())
        return self._eps_minmax

    @property
    def epsilon_3d_max(self):
        return float(self._mm()[1])

    @property
    def epsilon_3d_min(self):
        return float(self._mm()[0])

    def epsilon_3d_zero(self):
        self._mm().zero_()

    def close_update_epsilon_3d(self):
        self.update_epsilon_3d = False

    def open_update_epsilon_3d(self):
        self.update_epsilon_3d = True

    def print_epsilon(self):
        print("epsilon_3d_min: ", self.epsilon_3d_min)
        print("epsilon_3d_max: ", self.epsilon_3d_max)

    def forward(self, spatial_rgb, weight_and_index_list, ori_img, zero_init_mask: bool = False):
        ori_img = _lib.f32c(torch.as_tensor(ori_img), _cuda())           # GN:55
        x, x_rgba = gauss_gather(spatial_rgb, weight_and_index_list, ori_img, self.epsilon,
                                 self._mm() if self.update_epsilon_3d else None)
        # ---- cold tail, GN:121-157 (stock PyTorch)
        cla_x = x_rgba.transpose(2, 3).transpose(1, 2)
        cla_ori_img = ori_img.transpose(2, 3).transpose(1, 2)
        cla_x_3channel = torch.where(cla_x[:, 3:4] > 0, cla_x[:, :3], torch.full_like(cla_x[:, :3], 255.))
        cla_ori_img_3channel = torch.where(cla_ori_img[:, 3:4] > 0, cla_ori_img[:, :3],
                                           torch.full_like(cla_ori_img[:, :3], 255.))
        if self.model_name == "my_model":
            pass
        elif self.model_name == "vit_b_16":
            cla_x_3channel = self._resize(cla_x_3channel, 224)
            cla_ori_img_3channel = self._resize(cla_ori_img_3channel, 224)
        else:
            cla_x_3channel = self._resize(cla_x_3channel, 299)
            cla_ori_img_3channel = self._resize(cla_ori_img_3channel, 299)
        cla = self.model(cla_x_3channel)
        ori_cla = self.model(cla_ori_img_3channel)
        return x, x_rgba, cla, ori_img, ori_cla


class create_gauss_w(nn.Module):
    """GN:161-186: distances -> normalised Gaussian weights; returns (cat([w, idx], 1), dist)."""

    def __init__(self, device, c):
        super(create_gauss_w, self).__init__()
        self.top_number = 8
        self.device = device
        self.c = c

    def forward(self, dist_and_index_list):
        dai = _lib.f32c(dist_and_index_list, _cuda())
        if dai.dim() != 5 or dai.shape[1] != 2 or dai.shape[4] != 8:
            raise ValueError('dist_and_index_list must be [B,2,H,W,8] (CI:148-163)')
        B, P = dai.shape[0], dai.shape[2] * dai.shape[3]
        out = torch.empty_like(dai)
        _lib.check(_lib.load().nerfail_gauss_weight(_lib.dev(dai), B, P, float(self.c), _lib.dev(out), _lib.stream()))
        return out, dai[:, 0:1]
