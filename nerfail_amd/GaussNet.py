"""MI355X mirror of model/GaussNet.py (reference = GN): the differentiable pixel <-> 3-D-point map.

gauss_net.forward keeps its signature and 5-tuple return. The hot part (GN:53-119: 8-NN gather, weighted
sum, alpha, epsilon clip, composite onto the image) is one HIP kernel with a hand-written backward
(scatter-add with float atomics); the cold tail (GN:121-157: layout change, white background, Resize,
classifier) stays stock PyTorch, as SURVEY.md section 8(a16) scopes it.
"""
import ctypes
import weakref

import torch
from torch import nn

from . import _lib
from . import ops  # noqa: F401  (registers torch.ops.nerfail_mi.*)
from .run_nerf_helpers import _cuda


class GaussCSR:
    """Inverted index of a batch of views' 8-NN maps (built once per batch tensor, reused by every backward):
    row_ptr [Ns+1] int32, contrib / w_sorted / row_of [B*P*8] (nerfail_gauss_csr_build: the entries sorted by
    destination row, zero-weight entries dropped behind row_ptr[Ns])."""

    def __init__(self, wi, Ns):
        lib = _lib.load()
        B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
        nbytes = lib.nerfail_gauss_csr_workspace_bytes(Ns, B, P)
        if nbytes == 0:
            raise ValueError('batch too large for the 32-bit inverted index (B*P*8 < 2^31)')
        dev = wi.device
        self.Ns, self.B, self.P = Ns, B, P
        self.row_ptr = torch.empty((Ns + 1,), dtype=torch.int32, device=dev)
        self.contrib = torch.empty((B * P * 8,), dtype=torch.int32, device=dev)
        self.w_sorted = torch.empty((B * P * 8,), dtype=torch.float32, device=dev)
        self.row_of = torch.empty((B * P * 8,), dtype=torch.int32, device=dev)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _lib.check(lib.nerfail_gauss_csr_build(_lib.dev(wi), Ns, B, P, _lib.dev(self.row_ptr), _lib.dev(self.contrib),
                                               _lib.dev(self.w_sorted), _lib.dev(self.row_of), _lib.dev(ws), nbytes,
                                               _lib.stream()))
        self._wi = wi          # keeps the map alive so its address cannot be recycled under the cache key


_CSR_CACHE = {}
CSR_CACHE_SIZE = 8


def csr_for(wi, Ns):
    """Inverted index of a whole BATCH tensor, identity-keyed (address, version): for callers that keep one resident batch
    tensor and reuse it (tests, the multi-RHS kernel checks). The attack loop goes through view_indices() instead."""
    key = (wi.data_ptr(), wi._version, tuple(wi.shape), int(Ns))
    hit = _CSR_CACHE.pop(key, None)
    if hit is None:
        hit = GaussCSR(wi, Ns)
        while len(_CSR_CACHE) >= CSR_CACHE_SIZE:
            _CSR_CACHE.pop(next(iter(_CSR_CACHE)))
    _CSR_CACHE[key] = hit
    return hit


# ---------------------------------------------------------------------------------------------- per-view inverted indices
class ViewIndex:
    """Inverted index of ONE view's 8-NN map ("per-view CSR"): for every row of the perturbation
    table the (pixel, weight) pairs that gather from it, sorted by row, zero-weight pairs dropped. A view's map never
    changes, so this is built once per view - whatever batch the view later appears in (the reference's DataLoader
    shuffles: batch compositions do not repeat) - and can be stored next to index_and_weight/<split>/<i>.pth.
    Arrays are trimmed to the entries that exist (background pixels contribute none: ~40 % of 8*H*W on a real view)."""

    ARRAYS = ('packed', 'w_sorted', 'chunk_ord', 'pos')

    def __init__(self, wi_view=None, Ns=None, state=None):
        if state is not None:
            self.Ns, self.P = int(state['Ns']), int(state['P'])
            self.n_entries, self.n_rows = int(state['n_entries']), int(state['n_rows'])
            dev = _cuda()
            for k in self.ARRAYS:
                setattr(self, k, state[k].to(dev).contiguous())
            return
        lib = _lib.load()
        full = GaussCSR(wi_view.unsqueeze(0) if wi_view.dim() == 4 else wi_view, Ns)     # sorted entries (B = 1), temporary
        dev = full.row_ptr.device
        cap = full.contrib.numel()
        # compact form (nerfail_gauss_view_pack): 8 bytes per entry (pixel * 2 + row-start flag, weight), one row ordinal
        # per 512 entries, and pos[Ns] (row -> ordinal in the view's compact row-sum array, -1 = no entry)
        self.pos = torch.empty((Ns,), dtype=torch.int32, device=dev)
        packed = torch.empty((cap,), dtype=torch.int32, device=dev)
        chunk_ord = torch.zeros((int(lib.nerfail_gauss_view_chunks(cap)),), dtype=torch.int32, device=dev)
        counts = torch.empty((1,), dtype=torch.int32, device=dev)
        nb = lib.nerfail_gauss_view_pack_workspace_bytes(Ns)
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
        _lib.check(lib.nerfail_gauss_view_pack(_lib.dev(full.row_ptr), _lib.dev(full.row_of), _lib.dev(full.contrib), Ns, cap,
                                               _lib.dev(self.pos), _lib.dev(packed), _lib.dev(chunk_ord), _lib.dev(counts),
                                               _lib.dev(ws), nb, _lib.stream()))
        n = int(full.row_ptr[-1])                      # host reads: once per view, at build time only
        self.n_entries, self.n_rows = n, int(counts[0])
        self.Ns, self.P = Ns, full.P
        # trimmed to the entries that exist (kept >= 1 long so that the pointers stay valid for an all-background view)
        self.packed, self.w_sorted = packed[:max(n, 1)].clone(), full.w_sorted[:max(n, 1)].clone()
        self.chunk_ord = chunk_ord[:max(int(lib.nerfail_gauss_view_chunks(n)), 1)].clone()

    def nbytes(self):
        return sum(getattr(self, k).numel() * getattr(self, k).element_size() for k in self.ARRAYS)

    def fill(self, st):
        """Writes this index into a nerfail_view_index struct (the tensors must stay alive while it is used)."""
        for k in self.ARRAYS:
            setattr(st, k, getattr(self, k).data_ptr())
        st.n_entries, st.n_rows = self.n_entries, self.n_rows

    def state_dict(self):
        d = {k: getattr(self, k).cpu() for k in self.ARRAYS}
        d.update(Ns=self.Ns, P=self.P, n_entries=self.n_entries, n_rows=self.n_rows)
        return d

    def save(self, path):
        torch.save(self.state_dict(), path)

    @staticmethod
    def load(path):
        return ViewIndex(state=torch.load(path, map_location='cpu'))


def view_table(indices):
    """Host table of nerfail_view_index structs for nerfail_gauss_bwd_views, and the floats of scratch that call needs."""
    table = (_lib.ViewIndexStruct * len(indices))()
    for b, vi in enumerate(indices):
        vi.fill(table[b])
    return table, _lib.load().nerfail_gauss_bwd_views_scratch_floats(table, len(indices), indices[0].P, 1)


_VIEW_CACHE = {}                     # key -> ViewIndex, insertion order = LRU order
VIEW_CACHE_BYTES = 64 << 30          # 400 views of a scene are ~12 GB; HBM is 288 GB
_BATCH_KEYS = {}                     # (address, version, shape) of a resident batch tensor -> its views' keys


def fingerprints(wi):
    """Content keys of the views of a batch tensor [B,2,H,W,8] (one kernel + one 16*B-byte read back)."""
    B = wi.shape[0]
    out = torch.empty((B, 2), dtype=torch.int64, device=wi.device)
    _lib.check(_lib.load().nerfail_fingerprint(_lib.dev(wi), wi[0].numel(), B, _lib.dev(out), _lib.stream()))
    return [('fp',) + tuple(int(v) for v in row) + tuple(wi.shape[1:]) for row in out.cpu().tolist()]


def view_indices(wi, Ns, view_ids=None):
    """The ViewIndex of every view of the batch tensor `wi`, built on first sight and cached per VIEW.

    Key of a view: `view_ids[b]` when the caller names its views (dataset indices: free), else a content fingerprint of
    the view's map, so that the anonymous, freshly collated tensors a DataLoader yields (new address every iteration,
    MyDataset.py:199-204) still find their index. A resident tensor that is passed again (same address and version) skips
    the fingerprint."""
    Ns = int(Ns)
    if view_ids is not None:
        keys = [('id', v if isinstance(v, (str, bytes, tuple)) else int(v), Ns) for v in view_ids]
        if len(keys) != wi.shape[0]:
            raise ValueError('view_ids must name every view of the batch')
    else:
        # identity shortcut: only for the very same tensor OBJECT (weak reference). An address alone proves nothing - the
        # allocator hands a freed batch tensor's memory to the next one, with other views in it.
        ident = (wi.data_ptr(), wi._version, tuple(wi.shape))
        hit = _BATCH_KEYS.get(ident)
        keys = hit[1] if hit is not None and hit[0]() is wi else None
        if keys is None:
            keys = [k + (Ns,) for k in fingerprints(wi)]
            if len(_BATCH_KEYS) >= 64:
                _BATCH_KEYS.clear()
            _BATCH_KEYS[ident] = (weakref.ref(wi), keys)
    out = []
    for b, key in enumerate(keys):
        vi = _VIEW_CACHE.pop(key, None)
        if vi is None:
            vi = ViewIndex(wi[b], Ns)
            total = sum(v.nbytes() for v in _VIEW_CACHE.values()) + vi.nbytes()
            while _VIEW_CACHE and total > VIEW_CACHE_BYTES:
                total -= _VIEW_CACHE.pop(next(iter(_VIEW_CACHE))).nbytes()
        _VIEW_CACHE[key] = vi
        out.append(vi)
    return out


def load_view_indices(map_dir, ids, Ns, save_missing=True):
    """Per-view indices stored next to the maps they belong to: for every i in `ids` loads
    `<map_dir>/<i>.idx.pth` (ViewIndex.save) or, if it is not there yet, builds it from `<map_dir>/<i>.pth` (the float32
    [2,H,W,8] weight / index map dist_to_weight writes, DW:95-97) and stores it. The indices are registered in the cache;
    returns the view ids to hand to gauss_net.forward / nerfail_s_step (`view_ids=`), which then neither fingerprint nor
    rebuild anything, whatever batches the DataLoader composes."""
    import os
    out = []
    for i in ids:
        vid = (os.path.abspath(map_dir), int(i))
        side = os.path.join(map_dir, '%d.idx.pth' % int(i))
        if os.path.exists(side):
            vi = ViewIndex.load(side)
            if vi.Ns != int(Ns):
                raise ValueError('%s was built for a table of %d rows, not %d' % (side, vi.Ns, int(Ns)))
        else:
            wi = _lib.f32c(torch.load(os.path.join(map_dir, '%d.pth' % int(i)), map_location='cpu'), _cuda())
            vi = ViewIndex(wi, int(Ns))
            if save_missing:
                vi.save(side)
        register_view_index(vid, vi, Ns)
        out.append(vid)
    return out


def register_view_index(view_id, index, Ns=None):
    """Put a ViewIndex loaded from disk (ViewIndex.load) into the cache under the caller's view id."""
    _VIEW_CACHE[('id', view_id if isinstance(view_id, (str, bytes, tuple)) else int(view_id), int(Ns or index.Ns))] = index


class _GaussGather(torch.autograd.Function):
    """x, x_rgba = f(spatial_rgb); d/d(spatial_rgb) by the hand-written backward. weight/index/ori carry no grad.

    deterministic=True : gather-reduce over the cached inverted index (no atomics, bitwise reproducible).
    deterministic=False: scatter with float atomics (nerfail_gauss_bwd; no setup, order-dependent last bits)."""

    @staticmethod
    def forward(ctx, spatial, wi, ori, epsilon, eps_minmax, deterministic, view_ids=None):
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
        ctx.deterministic = bool(deterministic)
        ctx.view_ids = view_ids
        return x, x_rgba

    @staticmethod
    def backward(ctx, grad_x, grad_x_rgba):
        wi, ori, x = ctx.saved_tensors
        lib = _lib.load()
        B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
        n = 1
        for d in ctx.s_shape[:-1]:
            n *= d
        gx = _lib.f32c(grad_x) if grad_x is not None else None
        gr = _lib.f32c(grad_x_rgba) if grad_x_rgba is not None else None
        if ctx.deterministic:
            # every view reduced over its own index, the views' row sums added in view order (fixed order: bitwise
            # reproducible whatever else is in the cache)
            gs = torch.empty((n, 4), dtype=torch.float32, device=x.device)
            vis = view_indices(wi, n, ctx.view_ids)
            table, floats = view_table(vis)
            scratch = torch.empty((floats,), dtype=torch.float32, device=x.device)
            _lib.check(lib.nerfail_gauss_bwd_views(_lib.dev(ori), _lib.dev(x), _lib.dev(gx), _lib.dev(gr), table, B, n, P,
                                                   ctx.eps, _lib.dev(scratch), _lib.dev(gs), _lib.stream()))
        else:
            gs = torch.zeros((n, 4), dtype=torch.float32, device=x.device)
            _lib.check(lib.nerfail_gauss_bwd(_lib.dev(wi), _lib.dev(ori), _lib.dev(x), _lib.dev(gx), _lib.dev(gr),
                                             n, B, P, ctx.eps, _lib.dev(gs), _lib.stream()))
        return gs.reshape(ctx.s_shape), None, None, None, None, None, None


def gauss_gather(spatial_rgb, weight_and_index_list, ori_img, epsilon=None, eps_minmax=None, deterministic=True, view_ids=None):
    """Functional form of the hot part: returns (x, x_rgba), differentiable w.r.t. spatial_rgb. `view_ids` (optional):
    stable names of the batch's views (dataset indices) - keys of the per-view inverted indices (see view_indices)."""
    dev = _cuda()
    wi = weight_and_index_list
    if not (isinstance(wi, torch.Tensor) and wi.is_cuda and wi.dtype == torch.float32 and wi.is_contiguous()):
        wi = _lib.f32c(wi, dev)          # (a tensor already resident keeps its identity -> inverted-index cache hit)
    ori = _lib.f32c(ori_img, dev)
    if wi.dim() != 5 or wi.shape[1] != 2 or wi.shape[4] != 8:
        raise ValueError('weight_and_index_list must be [B,2,H,W,8] (DW:95-97)')
    # the kernels index ori / x as float4[B*H*W] and the table as float4[Ns]: mismatched shapes would read out of bounds
    if tuple(ori.shape) != (wi.shape[0], wi.shape[2], wi.shape[3], 4):
        raise ValueError('ori_img must be [B,H,W,4] with the B, H, W of weight_and_index_list, got %s vs %s'
                         % (tuple(ori.shape), tuple(wi.shape)))
    if spatial_rgb.dim() < 2 or spatial_rgb.shape[-1] != 4 or spatial_rgb.numel() == 0:
        raise ValueError('spatial_rgb must be [..., 4] (BGRA rows of the point set), got %s' % (tuple(spatial_rgb.shape),))
    if spatial_rgb.device != dev:
        spatial_rgb = spatial_rgb.to(dev)
    return _GaussGather.apply(spatial_rgb, wi, ori, epsilon, eps_minmax, deterministic, view_ids)


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
        self.deterministic = True    # backward = gather-reduce over a cached inverted index (False: float atomics)
        # The reference re-runs the classifier on the unperturbed images in EVERY forward (GN:157) although they never
        # change during an attack. True: keep the logits per (image tensor, version) - SURVEY 8f N4. Off by default
        # (a classifier in train() mode, e.g. with dropout, is not a pure function of its input).
        self.cache_ori_cla = False
        self._ori_cla_cache = {}
        # The cold tail hands the classifier a NCHW *view* of NHWC data (GN:121-131). MIOpen has no fp32 solver for that
        # layout and falls back to naive_conv_* kernels (17 ms per 8-view forward of the 800x800 victim CNN, 73 % of an
        # attack iteration). True: same values, copied to packed NCHW first (the layout torchvision models are tuned for).
        self.classifier_input_contiguous = True
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

    def forward(self, spatial_rgb, weight_and_index_list, ori_img, zero_init_mask: bool = False, view_ids=None):
        ori_img = _lib.f32c(torch.as_tensor(ori_img), _cuda())           # GN:55
        if not (isinstance(weight_and_index_list, torch.Tensor) and weight_and_index_list.is_cuda
                and weight_and_index_list.dtype == torch.float32 and weight_and_index_list.is_contiguous()):
            weight_and_index_list = _lib.f32c(weight_and_index_list, _cuda())
        self._last_ori, self._last_wi = ori_img, weight_and_index_list   # for logit_gradients()
        self._last_view_ids = view_ids
        x, x_rgba = gauss_gather(spatial_rgb, weight_and_index_list, ori_img, self.epsilon,
                                 self._mm() if self.update_epsilon_3d else None, self.deterministic, view_ids)
        # ---- cold tail, GN:121-157 (stock PyTorch)
        cla_x = x_rgba.transpose(2, 3).transpose(1, 2)
        cla_ori_img = ori_img.transpose(2, 3).transpose(1, 2)
        cla_x_3channel = torch.where(cla_x[:, 3:4] > 0, cla_x[:, :3], torch.full_like(cla_x[:, :3], 255.))
        cla_ori_img_3channel = torch.where(cla_ori_img[:, 3:4] > 0, cla_ori_img[:, :3],
                                           torch.full_like(cla_ori_img[:, :3], 255.))
        if self.classifier_input_contiguous:
            cla_x_3channel = cla_x_3channel.contiguous()
            cla_ori_img_3channel = cla_ori_img_3channel.contiguous()
        if self.model_name == "my_model":
            pass
        elif self.model_name == "vit_b_16":
            cla_x_3channel = self._resize(cla_x_3channel, 224)
            cla_ori_img_3channel = self._resize(cla_ori_img_3channel, 224)
        else:
            cla_x_3channel = self._resize(cla_x_3channel, 299)
            cla_ori_img_3channel = self._resize(cla_ori_img_3channel, 299)
        cla = self.model(cla_x_3channel)
        if self.cache_ori_cla:
            key = (ori_img.data_ptr(), ori_img._version, tuple(ori_img.shape))
            ori_cla = self._ori_cla_cache.get(key)
            if ori_cla is None:
                if len(self._ori_cla_cache) >= 64:
                    self._ori_cla_cache.clear()
                with torch.no_grad():
                    ori_cla = self.model(cla_ori_img_3channel)
                self._ori_cla_cache[key] = ori_cla
                self._ori_cla_keep = getattr(self, '_ori_cla_keep', [])[-63:] + [ori_img]   # keeps data_ptr from being recycled
        else:
            ori_cla = self.model(cla_ori_img_3channel)
        return x, x_rgba, cla, ori_img, ori_cla


    # ---- all class-logit gradients of one forward in ONE pass over the inverted index (DeepFool)
    def logit_gradients(self, spatial_rgb, weight_and_index_list, x, x_rgba, cla, classes):
        """d cla[0, k] / d spatial_rgb for every k in `classes` (<= 8): tensor [len(classes), *spatial_rgb.shape].

        `x, x_rgba, cla` are what forward() just returned for this spatial_rgb (graph still alive). The classifier is
        differentiated down to x_rgba by stock PyTorch (one backward per class); the pixel<->3-D map - the part the reference pays len(classes) scatter passes for - is one
        nerfail_gauss_bwd_view_multi call. Each slice is bitwise what autograd through forward() returns."""
        if cla.shape[0] != 1:
            raise ValueError('logit_gradients differentiates one view at a time (deepfool runs at batch 1, AN:82)')
        classes = [int(k) for k in classes]
        C = len(classes)
        if not 1 <= C <= 8:
            raise ValueError('1..8 classes per call')
        sel = torch.zeros((C, 1, cla.shape[1]), dtype=cla.dtype, device=cla.device)
        sel[torch.arange(C), 0, torch.tensor(classes)] = 1.0
        # classifier part (stock PyTorch / MIOpen): one backward per class. A single batched backward
        # (is_grads_batched=True) is available with self.batched_classifier_backward = True; on the 800x800 victim CNN
        # it measured slower (8.9 vs 7.9 ms for 8 classes), so it is off by default.
        if getattr(self, 'batched_classifier_backward', False):
            J = torch.autograd.grad(cla, x_rgba, grad_outputs=sel, retain_graph=True, is_grads_batched=True)[0]
        else:
            J = torch.stack([torch.autograd.grad(cla, x_rgba, grad_outputs=sel[i], retain_graph=True)[0] for i in range(C)])
        wi = weight_and_index_list
        B, P = wi.shape[0], wi.shape[2] * wi.shape[3]
        n = spatial_rgb.numel() // 4
        vi = view_indices(wi, n, getattr(self, '_last_view_ids', None))[0]      # one view (batch 1): its per-view index
        J = _lib.f32c(J).reshape(C, B * P, 4)
        ori = _lib.f32c(self._last_ori)
        out = torch.empty((C, n, 4), dtype=torch.float32, device=J.device)
        lib = _lib.load()
        st = _lib.ViewIndexStruct()
        vi.fill(st)
        scratch = torch.empty((lib.nerfail_gauss_bwd_views_scratch_floats(ctypes.byref(st), 1, P, C),), dtype=torch.float32, device=J.device)
        eps = -1.0 if self.epsilon is None else float(self.epsilon)
        x_c = _lib.f32c(x)            # bound to a name: see deepfool.py on pointers of temporaries
        _lib.check(lib.nerfail_gauss_bwd_view_multi(_lib.dev(ori), _lib.dev(x_c), _lib.dev(J), C, ctypes.byref(st), n, P, eps,
                                                    _lib.dev(scratch), _lib.dev(out), _lib.stream()))
        return out.reshape((C,) + tuple(spatial_rgb.shape))


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
        if not float(self.c) > 0:
            raise _lib.NerfailError('create_gauss_w: c must be positive')
        return torch.ops.nerfail_mi.gauss_weight(dai, float(self.c)), dai[:, 0:1]          # K9 as a registered op
