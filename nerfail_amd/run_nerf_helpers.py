"""MI355X mirror of Create_spatial_point_set/nerf_pytorch/run_nerf_helpers.py (reference = RH).

Same names, argument meaning and return structure as the reference; every numeric path runs in
libnerfail_hip.so (HIP, gfx950). No CPU fallback: CPU tensors are moved to the GPU, results live there.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib

# Misc (RH:9-11) - host-side one-liners, kept for drop-in completeness
class _Img2Mse(torch.autograd.Function):
    """mean((x - y)^2) and its gradient w.r.t. x from ONE launch (nerfail_mse); the backward is one multiply."""

    @staticmethod
    def forward(ctx, x, y):
        xc, yc = _lib.f32c(x), _lib.f32c(y)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(xc) if x.requires_grad else None
        _lib.check(_lib.load().nerfail_mse(_lib.dev(xc), _lib.dev(yc), xc.numel(), _lib.dev(loss), _lib.dev(dx), _lib.stream()))
        ctx.dx = dx
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.dx is None else ctx.dx * g), None


# the fused kernel is ONE workgroup (fixed-order tree, bitwise reproducible): right for a training batch (1024 rays x 3 = 3072
# values: 3 sweeps per thread), a serial crawl for an image - anything larger takes torch's multi-workgroup reduction
MSE_FUSED_MAX = 1 << 16


def img2mse(x, y):
    """RH:9. On the MI355X (float32 HIP tensors of equal shape, target without gradient) one fused kernel; any other
    input takes the reference's expression."""
    if (isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.is_cuda and y.is_cuda and x.dtype == torch.float32
            and y.dtype == torch.float32 and x.shape == y.shape and not y.requires_grad and 0 < x.numel() <= MSE_FUSED_MAX):
        return _Img2Mse.apply(x, y)
    return torch.mean((x - y) ** 2)


mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.Tensor([10.]).to(x.device))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _cuda():
    if not torch.cuda.is_available():
        raise RuntimeError('nerfail_amd needs an MI355X (torch.cuda.is_available() is False); no CPU path exists')
    return torch.device('cuda', torch.cuda.current_device())


# ----------------------------------------------------------------------------- positional encoding
class Embedder:
    """RH:15-50 with include_input=True, log_sampling=True, periodic_fns=[sin, cos] (get_embedder's kwargs)."""

    def __init__(self, multires):
        self.multires = int(multires)
        self.out_dim = 3 + 6 * self.multires

    def embed(self, inputs):
        x = _lib.f32c(inputs, _cuda())
        assert x.shape[-1] == 3, 'input_dims is 3 (RH:58)'
        flat = x.reshape(-1, 3)
        out = torch.empty((flat.shape[0], self.out_dim), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().nerfail_embed(_lib.dev(flat, 'inputs'), flat.shape[0], self.multires,
                                             _lib.dev(out), _lib.stream()))
        return out.reshape(x.shape[:-1] + (self.out_dim,))

    __call__ = embed


def get_embedder(multires, i=0, device=None):
    """RH:52-67. Returns (embed_fn, out_dim); embed_fn carries .multires so run_network can fuse it."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(multires)
    return eo, eo.out_dim


# ----------------------------------------------------------------------------- model
class NeRF(nn.Module):
    """RH:71-123: identical constructor, parameter names (state_dict keys) and forward contract.

    forward(x) takes the embedded batch [M, input_ch + input_ch_views] like the reference and runs the
    whole MLP in one HIP kernel. The render path does not go through forward(): run_network() hands
    raw points to the fused encode+MLP kernel (nerfail_mlp_fwd) using packed()."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super(NeRF, self).__init__()
        self.D = D
        self.W = W
        self.input_ch = input_ch
        self.input_ch_views = input_ch_views
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                        for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._packed = None
        self._packed_key = None
        self._packed16 = None
        self._packed16_key = None
        # 'f32' : exact-f32 MFMA kernel (default).  'f16x3': split-precision fp16 MFMA kernel, fp32-equivalent
        # results (each product as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi with fp32 accumulation), inference only.
        self.precision = 'f32'

    # -- MFMA-fragment-ordered weight image, rebuilt when any parameter changes (data_ptr, _version)
    def _skip(self):
        s = [i for i in self.skips if i < self.D - 1]
        if len(s) > 1:
            raise NotImplementedError('one skip connection supported (the reference always uses skips=[4])')
        return s[0] if s else -1

    def packed(self):
        if not self.use_viewdirs:
            raise NotImplementedError('HIP path implements use_viewdirs=True (all NeRFail configs, configs/*.txt)')
        params = list(self.parameters())
        if not params[0].is_cuda:
            raise RuntimeError('NeRF parameters are on %s: call .cuda() - there is no CPU path' % params[0].device)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        lib = _lib.load()
        n = lib.nerfail_mlp_packed_floats(self.D, self.W, self._skip())
        if n == 0:
            raise NotImplementedError('unsupported NeRF shape D=%d W=%d (W in {64,128,256})' % (self.D, self.W))
        mp = _lib.MlpParams()
        mp.D, mp.W, mp.input_ch, mp.input_ch_views, mp.skip = self.D, self.W, self.input_ch, self.input_ch_views, self._skip()
        keep = []

        def ptr(t):
            t = _lib.f32c(t)
            keep.append(t)
            return t.data_ptr()
        for i, l in enumerate(self.pts_linears):
            mp.pts_w[i] = ptr(l.weight)
            mp.pts_b[i] = ptr(l.bias)
        mp.views_w, mp.views_b = ptr(self.views_linears[0].weight), ptr(self.views_linears[0].bias)
        mp.feature_w, mp.feature_b = ptr(self.feature_linear.weight), ptr(self.feature_linear.bias)
        mp.alpha_w, mp.alpha_b = ptr(self.alpha_linear.weight), ptr(self.alpha_linear.bias)
        mp.rgb_w, mp.rgb_b = ptr(self.rgb_linear.weight), ptr(self.rgb_linear.bias)
        buf = torch.empty((n,), dtype=torch.float32, device=params[0].device)
        _lib.check(lib.nerfail_mlp_pack(mp, _lib.dev(buf), _lib.stream()))
        self._packed, self._packed_key = buf, key
        return buf

    def _mlp_params(self, keep):
        mp = _lib.MlpParams()
        mp.D, mp.W, mp.input_ch, mp.input_ch_views, mp.skip = self.D, self.W, self.input_ch, self.input_ch_views, self._skip()

        def ptr(t):
            t = _lib.f32c(t)
            keep.append(t)
            return t.data_ptr()
        for i, l in enumerate(self.pts_linears):
            mp.pts_w[i] = ptr(l.weight)
            mp.pts_b[i] = ptr(l.bias)
        mp.views_w, mp.views_b = ptr(self.views_linears[0].weight), ptr(self.views_linears[0].bias)
        mp.feature_w, mp.feature_b = ptr(self.feature_linear.weight), ptr(self.feature_linear.bias)
        mp.alpha_w, mp.alpha_b = ptr(self.alpha_linear.weight), ptr(self.alpha_linear.bias)
        mp.rgb_w, mp.rgb_b = ptr(self.rgb_linear.weight), ptr(self.rgb_linear.bias)
        return mp

    def packed_f16(self):
        """fp16 hi/lo weight image of the split-precision kernel (nerfail_mlp_pack_f16), cached like packed()."""
        params = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        self._f16_poll()                                  # an earlier pack's range check, if its value has arrived
        if self._packed16 is not None and key == self._packed16_key:
            return self._packed16
        lib = _lib.load()
        n = lib.nerfail_mlp_f16_image_bytes(self.D, self.W, self._skip())
        if n == 0:
            raise NotImplementedError('unsupported NeRF shape D=%d W=%d' % (self.D, self.W))
        self.check_f16x3_range()
        keep = []
        mp = self._mlp_params(keep)
        buf = torch.empty((n,), dtype=torch.uint8, device=params[0].device)
        _lib.check(lib.nerfail_mlp_pack_f16(mp, _lib.dev(buf), _lib.stream()))
        self._packed16, self._packed16_key = buf, key
        return buf

    F16X3_MAX_WEIGHT = 60.0       # the split-precision image holds fp16(w * 2^10): |w| * 1024 must stay below 65504

    def check_f16x3_range(self, sync=False):
        """The 'f16x3' kernels pre-scale the weights by 2^10 before the fp16 hi/lo split: a weight of magnitude >= ~64 would
        become inf (and its lo part NaN) - silently wrong results in a mode advertised as fp32-equivalent. EVERY pack
        measures max |weight| on the device (three small kernels). The first pack of a model waits for the value; later ones
        (training re-packs after every optimizer step) copy it to pinned host memory without stalling the stream and the
        NEXT call that touches the split-precision image (`packed_f16`, the transposed image, this method) looks at it: a
        weight that leaves the range is reported one forward later at the latest, whatever changed it (optimizer step,
        load_state_dict, a manual edit). sync=True waits for the value of this call. Activations are bounded by the same
        mechanism only through the weights: NeRF activations are O(1..100), fp16 range is 65504."""
        self._f16_poll()
        ws = [p.detach() for n_, p in self.named_parameters() if n_.endswith('weight')]
        with torch.no_grad():
            m = torch.stack(torch._foreach_norm(ws, float('inf'))).max()
        if sync or not m.is_cuda or not getattr(self, '_f16_checked_once', False):
            self._f16_checked_once = True
            self._f16_verdict(float(m))
            return
        # a measurement still on its way is WAITED for before its slot is reused: no verdict is ever overwritten unread
        # (every pack is checked, one forward later at the latest - the training loop packs once per step)
        self._f16_poll(wait=True)
        host = getattr(self, '_f16_host', None)
        if host is None:
            host = self._f16_host = torch.empty((1,), dtype=torch.float32).pin_memory()
        host.copy_(m.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._f16_pending = ev

    def _f16_poll(self, wait=False):
        """Looks at the max |weight| an earlier pack measured, if it has arrived (or waits for it)."""
        ev = getattr(self, '_f16_pending', None)
        if ev is None:
            return
        if wait:
            ev.synchronize()
        if ev.query():
            self._f16_pending = None
            self._f16_verdict(float(self._f16_host[0]))

    def _f16_verdict(self, m):
        if not m < self.F16X3_MAX_WEIGHT:          # (also catches NaN)
            raise ValueError("NeRF.precision = 'f16x3': max |weight| = %g is outside the range of the split-precision kernels "
                             "(|w| < %g); use precision = 'f32'" % (m, self.F16X3_MAX_WEIGHT))

    def forward(self, x):
        x = _lib.f32c(x, _cuda())
        assert x.shape[-1] == self.input_ch + self.input_ch_views
        flat = x.reshape(-1, x.shape[-1])
        out = torch.empty((flat.shape[0], 4), dtype=torch.float32, device=flat.device)
        _lib.check(_lib.load().nerfail_mlp_fwd_embedded(_lib.dev(self.packed()), self.D, self.W, self._skip(),
                                                        _lib.dev(flat, 'x'), flat.shape[0], _lib.dev(out), _lib.stream()))
        return out.reshape(x.shape[:-1] + (4,))


# ----------------------------------------------------------------------------- rays
def _k4(K):
    return _lib.host_floats([K[0][0], K[1][1], K[0][2], K[1][2]])


def _c2w12(c2w):
    c = torch.as_tensor(c2w).detach().to('cpu', torch.float32)[:3, :4].reshape(-1).tolist()
    return _lib.host_floats(c)


def get_rays(H, W, K, c2w):
    """RH:157-166 -> rays_o [H,W,3], rays_d [H,W,3] on the GPU."""
    dev = _cuda()
    rays_o = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    rays_d = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().nerfail_get_rays(int(H), int(W), _k4(K), _c2w12(c2w), _lib.dev(rays_o), _lib.dev(rays_d),
                                            _lib.stream()))
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """RH:169-176 (numpy twin used by the batching loader): same kernel, copied back to the host."""
    o, d = get_rays(H, W, K, c2w)
    return o.cpu().numpy(), d.cpu().numpy()


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    raise NotImplementedError('ndc_rays (RH:179-196) is LLFF-only; every NeRFail config sets ndc=False (RN:250-252)')


# ----------------------------------------------------------------------------- hierarchical sampling
_LINSPACE = {}


def linspace01(n, device):
    """torch.linspace(0,1,n) computed by the CPU kernel (the reference's bits), cached on the device."""
    key = (int(n), str(device))
    if key not in _LINSPACE:
        _LINSPACE[key] = torch.linspace(0., 1., steps=int(n)).to(device)
    return _LINSPACE[key]


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """RH:200-243. `u` (optional [R, N_samples]) supplies the uniform draws explicitly ("identical seeds")."""
    dev = _cuda()
    bins = _lib.f32c(bins, dev)
    weights = _lib.f32c(weights, dev)
    lead = bins.shape[:-1]
    nb = bins.shape[-1]
    assert weights.shape[-1] == nb - 1
    b2, w2 = bins.reshape(-1, nb), weights.reshape(-1, nb - 1)
    R = b2.shape[0]
    if pytest:                                             # RH:215-223
        np.random.seed(0)
        if det:
            u = torch.Tensor(np.linspace(0., 1., N_samples)).to(dev)
        else:
            u = torch.Tensor(np.random.rand(*(list(lead) + [N_samples]))).to(dev)
    elif u is None:
        u = linspace01(N_samples, dev) if det else torch.rand(list(lead) + [N_samples], device=dev)
    u = _lib.f32c(u, dev)
    is_row = int(u.dim() == 1)
    if not is_row:
        u = u.reshape(R, N_samples)
    from . import ops  # noqa: F401
    out = torch.ops.nerfail_mi.sample_pdf(b2, w2, u)                                        # K6 as a registered op
    return out.reshape(tuple(lead) + (N_samples,))
