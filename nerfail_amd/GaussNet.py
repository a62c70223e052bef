"""MI355X mirror of model/GaussNet.py (reference = GN): the differentiable pixel <-> 3-D-point map.

gauss_net.forward keeps its signature and 5-tuple return. The hot part (GN:53-119: 8-NN gather, weighted
sum, alpha, epsilon clip, composite onto the image) is one HIP kernel with a hand-written backward
(scatter-add with float atomics); the cold tail (GN:121-157: layout change, white background, Resize,
classifier) stays stock PyTorch, as SURVEY.md section 8(a16) scopes it.
"""
import ctypes
import os
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
        self.verified = False          # set once the index has been checked against (or built from) a map seen in this process
        if state is not None:
            self.Ns, self.P = int(state['Ns']), int(state['P'])
            self.n_entries, self.n_rows = int(state['n_entries']), int(state['n_rows'])
            self.fp = tuple(state['fp']) if state.get('fp') is not None else None
            self.src = state.get('src')                  # (mtime_ns, size) of the map file the index was built from
            dev = _cuda()
            for k in self.ARRAYS:
                setattr(self, k, state[k].to(dev).contiguous())
            return
        lib = _lib.load()
        self.fp = fingerprints(wi_view.unsqueeze(0) if wi_view.dim() == 4 else wi_view)[0]      # content key of the map
        self.src = None
        self.verified = True
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
        d.update(Ns=self.Ns, P=self.P, n_entries=self.n_entries, n_rows=self.n_rows, fp=self.fp, src=self.src)
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
        if vi.P != indices[0].P or vi.Ns != indices[0].Ns:
            raise ValueError('view indices of different image / table sizes in one batch: P %d vs %d, Ns %d vs %d'
                             % (vi.P, indices[0].P, vi.Ns, indices[0].Ns))
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


def _view_key(v, Ns):
    return ('id', v if isinstance(v, (str, bytes, tuple)) else int(v), int(Ns))


def view_indices(wi, Ns, view_ids=None):
    """The ViewIndex of every view of the batch `wi` (a tensor [B,2,H,W,8] or a list of per-view tensors [2,H,W,8]), built
    on first sight and cached per VIEW.

    Key of a view: `view_ids[b]` when the caller names its views, else a content fingerprint of the view's map, so that the
    anonymous, freshly collated tensors a DataLoader yields (new address every iteration, MyDataset.py:199-204) still find
    their index. A resident tensor that is passed again (same address and version) skips the fingerprint.

    A caller-named id is only as good as the caller's naming: ids must be unique per (scene, split, resolution) - use
    tuples like (map_dir, i), as load_view_indices / load_view_maps do; a bare int reused for another split would name
    another view's index. What IS checked: image size and table size on every use (a mismatch raises instead of reading
    out of bounds), and the index's stored fingerprint against the map the first time an id is used with a map at hand."""
    Ns = int(Ns)
    views = list(wi) if isinstance(wi, (list, tuple)) else [wi[b] for b in range(wi.shape[0])]
    if view_ids is not None:
        keys = [_view_key(v, Ns) for v in view_ids]
        if len(keys) != len(views):
            raise ValueError('view_ids must name every view of the batch')
    else:
        if isinstance(wi, torch.Tensor):
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
        else:
            keys = [fingerprints(v.unsqueeze(0))[0] + (Ns,) for v in views]
    out = []
    for b, key in enumerate(keys):
        wv = views[b]
        P = int(wv.shape[-3] * wv.shape[-2]) if wv is not None else None
        vi = _VIEW_CACHE.pop(key, None)
        if vi is None:
            if wv is None:
                raise KeyError('view %r has neither a cached index nor a map to build one from' % (key,))
            vi = ViewIndex(wv, Ns)
            total = sum(v.nbytes() for v in _VIEW_CACHE.values()) + vi.nbytes()
            while _VIEW_CACHE and total > VIEW_CACHE_BYTES:
                total -= _VIEW_CACHE.pop(next(iter(_VIEW_CACHE))).nbytes()
        else:
            if vi.Ns != Ns or (P is not None and vi.P != P):
                _VIEW_CACHE[key] = vi
                raise ValueError('cached index of view %r was built for %d pixels / %d table rows, this call has %s / %d: the id '
                                 'names another view (ids must be unique per scene, split and resolution)' % (key, vi.P, vi.Ns, P, Ns))
            if key[0] == 'id' and not vi.verified and wv is not None and wv.is_cuda:
                fp = fingerprints(wv.unsqueeze(0))[0]        # once per id: one 0.1 ms kernel
                if vi.fp is not None and tuple(vi.fp) != tuple(fp):
                    _VIEW_CACHE[key] = vi
                    raise ValueError('the index registered for view %r was built from a different map (fingerprint mismatch): '
                                     'stale sidecar or a reused id' % (key,))
                vi.fp, vi.verified = fp, True
        _VIEW_CACHE[key] = vi
        out.append(vi)
    return out


def _file_sig(path):
    st = os.stat(path)
    return (int(st.st_mtime_ns), int(st.st_size))


def load_view_indices(map_dir, ids, Ns, save_missing=True):
    """Per-view indices stored next to the maps they belong to: for every i in `ids` loads
    `<map_dir>/<i>.idx.pth` (ViewIndex.save) or, if it is not there yet - or was built from another version of the map
    (the sidecar records the map file's mtime and size, and the map's fingerprint) - builds it from `<map_dir>/<i>.pth`
    (the float32 [2,H,W,8] weight / index map dist_to_weight writes, DW:95-97) and stores it. The indices are registered
    in the cache; returns the view ids to hand to gauss_net.forward / nerfail_s_step (`view_ids=`), which then neither
    fingerprint nor rebuild anything, whatever batches the DataLoader composes."""
    out = []
    for i in ids:
        vid = (os.path.abspath(map_dir), int(i))
        side = os.path.join(map_dir, '%d.idx.pth' % int(i))
        mpath = os.path.join(map_dir, '%d.pth' % int(i))
        vi = None
        if os.path.exists(side):
            vi = ViewIndex.load(side)
            if vi.Ns != int(Ns):
                raise ValueError('%s was built for a table of %d rows, not %d' % (side, vi.Ns, int(Ns)))
            if os.path.exists(mpath) and (vi.src is None or tuple(vi.src) != _file_sig(mpath)):
                vi = None                   # the map was regenerated (another c, another NeRF, another resolution): rebuild
        if vi is None:
            wi = _lib.f32c(torch.load(mpath, map_location='cpu'), _cuda())
            vi = ViewIndex(wi, int(Ns))
            vi.src = _file_sig(mpath)
            if save_missing:
                vi.save(side)
        register_view_index(vid, vi, Ns, replace=True)
        out.append(vid)
    return out


def register_view_index(view_id, index, Ns=None, replace=False):
    """Put a ViewIndex loaded from disk (ViewIndex.load) into the cache under the caller's view id. An id that is already
    taken by an index of ANOTHER map (different fingerprint or size) is refused unless replace=True."""
    key = _view_key(view_id, Ns or index.Ns)
    old = _VIEW_CACHE.get(key)
    if old is not None and not replace and old is not index and (
            old.P != index.P or (old.fp is not None and index.fp is not None and tuple(old.fp) != tuple(index.fp))):
        raise ValueError('view id %r already names the index of another map; pass replace=True to overwrite it' % (view_id,))
    _VIEW_CACHE[key] = index


# ---------------------------------------------------------------------------------------------- device-resident views
# The reference's dataset hands the attack loop a freshly loaded map per view and step (MyDataset.py:199-204: torch.load of
# 41 MB + a cv2.imread per view; AS:304-317 passes them straight to the net): 8 x 51 MB over PCIe per iteration would be
# ~20x the 0.35 ms gauss path. A view's map and image never change, so they are kept on the device BY VIEW ID (288 GB of
# HBM hold a whole 400-view scene: 16 GB of maps + 1 GB of uint8 images): a forward that names its views finds them here
# and ignores the tensors it was handed.
_VIEW_MAPS = {}                      # key -> [2,H,W,8] float32 device tensor, insertion order = LRU order
_VIEW_ORI = {}                       # key -> [H,W,4] uint8 (or float32) device tensor
_VIEW_ORI_EPOCH = {}                 # key -> how often the resident image of that id was (re)registered: part of the logit-cache key
_VIEW_PASSED_OK = set()              # keys whose resident map was compared once with a map the caller passed under that id
VIEW_MAPS_BYTES = 96 << 30


def _lru_put(store, key, t, budget):
    store.pop(key, None)
    total = sum(v.numel() * v.element_size() for v in store.values()) + t.numel() * t.element_size()
    while store and total > budget:
        k0 = next(iter(store))
        total -= store[k0].numel() * store[k0].element_size()
        del store[k0]
    store[key] = t


def register_view(view_id, Ns, weight_and_index=None, ori_img=None):
    """Keep a view's [2,H,W,8] map and / or its image (uint8 BGRA as the dataset reads it, or float) on the device under
    `view_id`. Also builds (or verifies) the view's inverted index."""
    key = _view_key(view_id, Ns)
    dev = _cuda()
    if weight_and_index is not None:
        wi = _lib.f32c(torch.as_tensor(weight_and_index), dev)
        if wi.dim() != 4 or wi.shape[0] != 2 or wi.shape[3] != 8:
            raise ValueError('a view map must be [2,H,W,8] (DW:95-97)')
        view_indices([wi], Ns, [view_id])
        _lru_put(_VIEW_MAPS, key, wi, VIEW_MAPS_BYTES)
        _VIEW_PASSED_OK.discard(key)
    if ori_img is not None:
        o = torch.as_tensor(ori_img)
        o = o.to(dev).contiguous() if o.dtype == torch.uint8 else _lib.f32c(o, dev)
        _lru_put(_VIEW_ORI, key, o, VIEW_MAPS_BYTES)
        _VIEW_ORI_EPOCH[key] = _VIEW_ORI_EPOCH.get(key, 0) + 1      # cached original-image logits of this id are stale now
    return key


def load_view_maps(map_dir, ids, Ns, images=None):
    """Device-resident maps for the attack loop: `<map_dir>/<i>.pth` of every i in `ids` is loaded ONCE,
    kept on the device under the same ids load_view_indices uses, its inverted index loaded / built alongside. `images`:
    optional {i: uint8 [H,W,4] array} (the cv2.imread result of MyDataset.py:200) kept resident too. Returns the view ids
    for gauss_net.forward / nerfail_s_step (`view_ids=`)."""
    vids = load_view_indices(map_dir, ids, Ns)
    for i, vid in zip(ids, vids):
        key = _view_key(vid, Ns)
        if key not in _VIEW_MAPS:
            wi = _lib.f32c(torch.load(os.path.join(map_dir, '%d.pth' % int(i)), map_location='cpu'), _cuda())
            vi = _VIEW_CACHE[key]
            if vi.P != wi.shape[1] * wi.shape[2]:
                raise ValueError('%s/%d.pth does not match its index (pixels %d vs %d)' % (map_dir, int(i), wi.shape[1] * wi.shape[2], vi.P))
            _lru_put(_VIEW_MAPS, key, wi, VIEW_MAPS_BYTES)
        if images is not None and i in images:
            register_view(vid, Ns, ori_img=images[i])
    return vids


class BatchViews:
    """The views of one forward, one entry per view: device map [2,H,W,8], device image ([H,W,4] uint8 or float32), key.
    Built by gauss_net.resolve_views from what the caller passed and what is resident."""

    def __init__(self, wi, ori, ori_u8, view_ids, Ns, batch_wi=None, batch_ori=None):
        self.wi, self.ori, self.ori_u8, self.view_ids, self.Ns = wi, ori, ori_u8, view_ids, Ns
        self.batch_wi, self.batch_ori = batch_wi, batch_ori           # the contiguous batch tensors, when they came that way
        self.ori_src = None          # identity of the image tensor the caller passed, when it lives on the device (logit cache key)
        self.B = len(wi)
        self.H, self.W = int(wi[0].shape[1]), int(wi[0].shape[2])
        self.P = self.H * self.W

    def table(self):
        t = (_lib.ViewFwdStruct * self.B)()
        for b in range(self.B):
            t[b].weight_and_index, t[b].ori_img = self.wi[b].data_ptr(), self.ori[b].data_ptr()
        return t

    def indices(self):
        return view_indices(self.batch_wi if (self.batch_wi is not None and self.view_ids is None) else self.wi, self.Ns, self.view_ids)

    def ori_float(self):
        """[B,H,W,4] float32 (GN:55), materialised only when somebody needs it."""
        if self.batch_ori is not None and self.batch_ori.dtype == torch.float32:
            return self.batch_ori
        return torch.stack([o.float() for o in self.ori])

    def wi_batch(self):
        return self.batch_wi if self.batch_wi is not None else torch.stack(self.wi)


def resolve_views(spatial_rgb, weight_and_index_list, ori_img, view_ids=None, keep_resident=False):
    """What the kernels will read for this batch. With `view_ids`, a view whose map / image is resident (load_view_maps,
    register_view, or an earlier call with keep_resident) is taken from the device and the passed tensor is NOT touched -
    the reference-shaped loop hands over CPU tensors from a DataLoader every iteration."""
    dev = _cuda()
    Ns = int(spatial_rgb.numel() // 4)
    wi_in, ori_in = weight_and_index_list, ori_img
    if isinstance(wi_in, (list, tuple)):
        # a batch handed over as per-view tensors (MyDataset.collate_views): nothing is stacked, ids travel with the list
        if view_ids is None and getattr(wi_in, 'view_ids', None) is not None:
            view_ids = wi_in.view_ids
        wl = [w if (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()) else _lib.f32c(w, dev) for w in wi_in]
        ol = [(o.to(dev).contiguous() if o.dtype == torch.uint8 else _lib.f32c(o, dev)) for o in ori_in]
        if len(wl) != len(ol) or (view_ids is not None and len(view_ids) != len(wl)):
            raise ValueError('maps, images and view ids of a batch must have the same length')
        H, W = wl[0].shape[1], wl[0].shape[2]
        for w, o in zip(wl, ol):
            if tuple(w.shape) != (2, H, W, 8) or tuple(o.shape) != (H, W, 4):
                raise ValueError('a view must be a [2,H,W,8] map with a [H,W,4] image, got %s and %s' % (tuple(w.shape), tuple(o.shape)))
        if len({o.dtype for o in ol}) > 1:
            ol = [o.float() for o in ol]
        return BatchViews(wl, ol, ol[0].dtype == torch.uint8, view_ids, Ns)
    B = len(view_ids) if view_ids is not None else wi_in.shape[0]
    keys = [_view_key(v, Ns) for v in view_ids] if view_ids is not None else [None] * B
    res_wi = [_VIEW_MAPS.get(k) if k is not None else None for k in keys]
    res_ori = [_VIEW_ORI.get(k) if k is not None else None for k in keys]
    batch_wi = batch_ori = None
    # A resident view wins over the tensor the caller passes under the same id (that is the point: the reference-shaped loop
    # hands over freshly loaded tensors every iteration). ONCE per id the passed map - if it is a device tensor, i.e. free to
    # look at - is compared with the resident one by content fingerprint, so that an id reused for another scene / split or a
    # regenerated map file raises instead of silently differentiating through the wrong view.
    if view_ids is not None and isinstance(wi_in, torch.Tensor) and wi_in.is_cuda and wi_in.dim() == 5 and wi_in.shape[0] == B:
        todo = [b for b, k in enumerate(keys) if res_wi[b] is not None and k not in _VIEW_PASSED_OK]
        if todo and wi_in.dtype == torch.float32 and wi_in.is_contiguous():
            got = fingerprints(wi_in)
            for b in todo:
                if tuple(wi_in.shape[1:]) != tuple(res_wi[b].shape) or got[b] != fingerprints(res_wi[b].unsqueeze(0))[0]:
                    raise ValueError('view id %r is resident with a DIFFERENT map than the one passed under that id (reused id or '
                                     'regenerated map file): register_view(..., weight_and_index=) to replace it' % (view_ids[b],))
                _VIEW_PASSED_OK.add(keys[b])
    if any(w is None for w in res_wi):
        if wi_in is None:
            raise KeyError('a view of the batch is not resident and no weight_and_index_list was passed')
        if not isinstance(wi_in, torch.Tensor) or wi_in.dim() != 5 or wi_in.shape[1] != 2 or wi_in.shape[4] != 8:
            raise ValueError('weight_and_index_list must be [B,2,H,W,8] (DW:95-97)')
        if all(w is None for w in res_wi):
            # (a tensor already resident keeps its identity -> inverted-index cache hit)
            batch_wi = wi_in if (wi_in.is_cuda and wi_in.dtype == torch.float32 and wi_in.is_contiguous()) else _lib.f32c(wi_in, dev)
            res_wi = [batch_wi[b] for b in range(B)]
        else:
            res_wi = [w if w is not None else _lib.f32c(wi_in[b], dev) for b, w in enumerate(res_wi)]
        if keep_resident and view_ids is not None:
            for b, k in enumerate(keys):
                if k not in _VIEW_MAPS:
                    _lru_put(_VIEW_MAPS, k, res_wi[b].clone() if batch_wi is not None else res_wi[b], VIEW_MAPS_BYTES)
    if any(o is None for o in res_ori):
        o_in = torch.as_tensor(ori_in)
        if o_in.dim() != 4 or o_in.shape[0] != B or o_in.shape[-1] != 4:
            raise ValueError('ori_img must be [B,H,W,4] with the B of the batch, got %s' % (tuple(o_in.shape),))
        if all(o is None for o in res_ori) and o_in.dtype != torch.uint8:
            batch_ori = _lib.f32c(o_in, dev)                                               # GN:55
            res_ori = [batch_ori[b] for b in range(B)]
        else:
            up = o_in.to(dev).contiguous() if o_in.dtype == torch.uint8 else _lib.f32c(o_in, dev)
            res_ori = [o if o is not None else up[b] for b, o in enumerate(res_ori)]
        if keep_resident and view_ids is not None:
            for b, k in enumerate(keys):
                if k not in _VIEW_ORI:
                    _lru_put(_VIEW_ORI, k, res_ori[b].clone(), VIEW_MAPS_BYTES)
    kinds = {o.dtype for o in res_ori}
    if len(kinds) > 1:                                       # mixed uint8 / float images: one format for the kernel
        res_ori = [o.float() for o in res_ori]
        batch_ori = None
    H, W = res_wi[0].shape[1], res_wi[0].shape[2]
    for w, o in zip(res_wi, res_ori):
        # the kernels index ori / x as [B*H*W] pixels and the table as float4[Ns]: mismatched shapes would read out of bounds
        if tuple(w.shape) != (2, H, W, 8) or tuple(o.shape) != (H, W, 4):
            raise ValueError('ori_img must be [B,H,W,4] with the B, H, W of weight_and_index_list, got %s vs %s' % (tuple(o.shape), tuple(w.shape)))
    bv = BatchViews(res_wi, res_ori, res_ori[0].dtype == torch.uint8, view_ids, Ns, batch_wi, batch_ori)
    if isinstance(ori_in, torch.Tensor) and ori_in.is_cuda:
        bv.ori_src = ('tensor', ori_in.data_ptr(), ori_in._version, tuple(ori_in.shape), str(ori_in.dtype))
        bv._ori_keep = ori_in        # the address cannot be recycled while the key may be looked up
    return bv


def hot_forward(spatial_rgb, views, epsilon=None, eps_minmax=None, need_x=True, need_aux=False):
    """K10 over a BatchViews: (x or None, x_rgba, (aux_alpha, aux_mask) or None); no autograd."""
    dev = _cuda()
    s = _lib.f32c(spatial_rgb, dev).reshape(-1, 4)
    B, H, W = views.B, views.H, views.W
    x = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev) if need_x else None
    x_rgba = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
    aux = (torch.empty((B, H, W), dtype=torch.float32, device=dev), torch.empty((B, H, W), dtype=torch.uint8, device=dev)) if need_aux else None
    eps = -1.0 if epsilon is None else float(epsilon)
    _lib.check(_lib.load().nerfail_gauss_fwd_views(_lib.dev(s, 'spatial_rgb'), s.shape[0], views.table(), B, views.P, int(views.ori_u8), eps,
                                                   _lib.dev(x), _lib.dev(x_rgba), _lib.dev(aux[0]) if aux else None,
                                                   _lib.dev(aux[1]) if aux else None, _lib.dev(eps_minmax), _lib.stream()))
    return x, x_rgba, aux


def hot_backward_rgb(aux, grad_x_rgba, views, out=None):
    """K11, rgb-gradient-only form: d/d(spatial rgb) as [Ns,3] (flat buffer `out` of >= 3 Ns floats, e.g. with a tail slot
    for the loss so that ONE all-reduce moves both)."""
    dev = grad_x_rgba.device
    Ns = views.Ns
    vis = views.indices()
    table, floats = view_table(vis)
    scratch = torch.empty((floats,), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty((3 * Ns,), dtype=torch.float32, device=dev)
    gr = _lib.f32c(grad_x_rgba)
    _lib.check(_lib.load().nerfail_gauss_bwd_views_rgb(_lib.dev(aux[0]), _lib.dev(aux[1]), _lib.dev(gr), table, views.B, Ns, views.P,
                                                       _lib.dev(scratch), _lib.dev(out), _lib.stream()))
    return out


def hot_backward_rgb_step(aux, grad_x_rgba, views, spatial, spatial_init, a, epsilon, targeted, grad_out=None):
    """K11 (rgb form) + K12 in one pass (nerfail_gauss_bwd_views_rgb_step, round 6): the new perturbation table after the sign step
    AS:352-392; the gradient itself is written only if `grad_out` (>= 3 Ns floats) is given. Bit-identical to hot_backward_rgb
    followed by attack.igsm_step_rgb."""
    dev = grad_x_rgba.device
    Ns = views.Ns
    table, floats = view_table(views.indices())
    scratch = torch.empty((floats,), dtype=torch.float32, device=dev)
    s, s0 = _lib.f32c(spatial, dev), _lib.f32c(spatial_init, dev)
    if s.numel() != 4 * Ns or s0.numel() != 4 * Ns:
        raise ValueError('spatial / spatial_init must hold Ns = %d rows of 4 floats' % Ns)
    if grad_out is None and views.B > 16:
        grad_out = torch.empty((3 * Ns,), dtype=torch.float32, device=dev)
    out = torch.empty_like(s)
    _lib.check(_lib.load().nerfail_gauss_bwd_views_rgb_step(_lib.dev(aux[0]), _lib.dev(aux[1]), _lib.dev(_lib.f32c(grad_x_rgba)), table, views.B,
                                                           Ns, views.P, _lib.dev(scratch), _lib.dev(grad_out), _lib.dev(s), _lib.dev(s0),
                                                           float(a), float(epsilon), int(bool(targeted)), _lib.dev(out), _lib.stream()))
    return out


class _GaussGather(torch.autograd.Function):
    """x, x_rgba = f(spatial_rgb); d/d(spatial_rgb) by the hand-written backward. The views carry no grad.

    deterministic=True : gather-reduce over the cached per-view inverted indices (no atomics, bitwise reproducible).
    deterministic=False: scatter with float atomics (nerfail_gauss_bwd; no setup, order-dependent last bits)."""

    @staticmethod
    def forward(ctx, spatial, views, epsilon, eps_minmax, deterministic):
        x, x_rgba, _ = hot_forward(spatial, views, epsilon, eps_minmax, need_x=True)
        ctx.save_for_backward(x)
        ctx.views = views
        ctx.eps = -1.0 if epsilon is None else float(epsilon)
        ctx.s_shape = tuple(spatial.shape)
        ctx.deterministic = bool(deterministic)
        return x, x_rgba

    @staticmethod
    def backward(ctx, grad_x, grad_x_rgba):
        (x,) = ctx.saved_tensors
        views = ctx.views
        lib = _lib.load()
        B, P = views.B, views.P
        n = 1
        for d in ctx.s_shape[:-1]:
            n *= d
        gx = _lib.f32c(grad_x) if grad_x is not None else None
        gr = _lib.f32c(grad_x_rgba) if grad_x_rgba is not None else None
        ori = views.ori_float()
        if ctx.deterministic:
            # every view reduced over its own index, the views' row sums added in view order (fixed order: bitwise
            # reproducible whatever else is in the cache)
            gs = torch.empty((n, 4), dtype=torch.float32, device=x.device)
            table, floats = view_table(views.indices())
            scratch = torch.empty((floats,), dtype=torch.float32, device=x.device)
            _lib.check(lib.nerfail_gauss_bwd_views(_lib.dev(ori), _lib.dev(x), _lib.dev(gx), _lib.dev(gr), table, B, n, P,
                                                   ctx.eps, _lib.dev(scratch), _lib.dev(gs), _lib.stream()))
        else:
            gs = torch.zeros((n, 4), dtype=torch.float32, device=x.device)
            _lib.check(lib.nerfail_gauss_bwd(_lib.dev(views.wi_batch()), _lib.dev(ori), _lib.dev(x), _lib.dev(gx), _lib.dev(gr),
                                             n, B, P, ctx.eps, _lib.dev(gs), _lib.stream()))
        return gs.reshape(ctx.s_shape), None, None, None, None


def gauss_gather(spatial_rgb, weight_and_index_list, ori_img, epsilon=None, eps_minmax=None, deterministic=True, view_ids=None):
    """Functional form of the hot part: returns (x, x_rgba), differentiable w.r.t. spatial_rgb. `view_ids` (optional):
    stable names of the batch's views (dataset indices) - keys of the per-view inverted indices (see view_indices) and of
    the device-resident maps / images (load_view_maps, register_view)."""
    dev = _cuda()
    if spatial_rgb.dim() < 2 or spatial_rgb.shape[-1] != 4 or spatial_rgb.numel() == 0:
        raise ValueError('spatial_rgb must be [..., 4] (BGRA rows of the point set), got %s' % (tuple(spatial_rgb.shape),))
    if spatial_rgb.device != dev:
        spatial_rgb = spatial_rgb.to(dev)
    views = resolve_views(spatial_rgb, weight_and_index_list, ori_img, view_ids)
    return _GaussGather.apply(spatial_rgb, views, epsilon, eps_minmax, deterministic)


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
        self.rgb_grad_only = True    # nerfail_s_step: rgb-gradient-only step path (False: full autograd path, four channels)
        self.fused_sign_step = True  # nerfail_s_step on ONE rank: the sign step as the epilogue of the gather backward (round 6)
        # The reference re-runs the classifier on the unperturbed images in EVERY forward (GN:157) although they never
        # change during an attack. None (default, round 6) = automatic: the logits are kept per set of VIEW
        # IDS (view_ids=: stable names of views whose map and image do not change) for as long as the classifier is a pure,
        # frozen function - every module in eval() mode (AS:281-282: model.train(False)), no parameter requiring grad
        # (AS:284-287) - and none of its parameters / buffers was written (their (address, version) pairs are part of the
        # key). True: also per image TENSOR (address, version) and whatever the classifier's mode. False: never.
        self.cache_ori_cla = None
        self._ori_cla_cache = {}
        self._ori_cla_model_key = None
        # The cold tail hands the classifier a NCHW *view* of NHWC data (GN:121-131). MIOpen has no fp32 solver for that
        # layout and falls back to naive_conv_* kernels (17 ms per 8-view forward of the 800x800 victim CNN, 73 % of an
        # attack iteration). True: same values, copied to packed NCHW first (the layout torchvision models are tuned for).
        self.classifier_input_contiguous = True
        # True: a view that is named (view_ids=) and not yet resident is kept on the device after its first upload - the
        # reference-shaped loop (CPU tensors from a DataLoader every iteration) then pays PCIe once per view, not per step
        self.keep_views_resident = False
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
        dev = _cuda()
        if spatial_rgb.device != dev:
            spatial_rgb = spatial_rgb.to(dev)
        views = resolve_views(spatial_rgb, weight_and_index_list, ori_img, view_ids, self.keep_views_resident)
        ori_img = views.ori_float()                                      # GN:55
        self._last_views = views                                         # for logit_gradients()
        self._last_ori = ori_img
        x, x_rgba = _GaussGather.apply(spatial_rgb, views, self.epsilon, self._mm() if self.update_epsilon_3d else None,
                                       self.deterministic)
        # the logits of the unperturbed images: keyed by view ids when the caller names its views (then cached by default,
        # see cache_ori_cla); without ids the key stays the float image tensor's identity, as before
        cla, ori_cla = self.cold_tail(x_rgba, ori_img, ori_key=self._ori_key(views) if views.view_ids is not None else None)
        return x, x_rgba, cla, ori_img, ori_cla

    def cold_tail(self, x_rgba, ori_img, ori_key=None):
        """GN:121-157 (stock PyTorch): NHWC -> NCHW, white background, Resize, the classifier on the perturbed and on the
        original images. `ori_img` may be a callable that yields the float images (only evaluated when their logits are
        not cached); `ori_key`: cache key of the original images' logits (default: the tensor's identity)."""
        cla_x = x_rgba.transpose(2, 3).transpose(1, 2)
        cla_x_3channel = torch.where(cla_x[:, 3:4] > 0, cla_x[:, :3], torch.full_like(cla_x[:, :3], 255.))
        if self.classifier_input_contiguous:
            cla_x_3channel = cla_x_3channel.contiguous()
        size = None if self.model_name == "my_model" else (224 if self.model_name == "vit_b_16" else 299)
        if size is not None:
            cla_x_3channel = self._resize(cla_x_3channel, size)
        cla = self.model(cla_x_3channel)

        def ori_logits(grad):
            o = ori_img() if callable(ori_img) else ori_img
            c = o.transpose(2, 3).transpose(1, 2)
            c3 = torch.where(c[:, 3:4] > 0, c[:, :3], torch.full_like(c[:, :3], 255.))
            if self.classifier_input_contiguous:
                c3 = c3.contiguous()
            if size is not None:
                c3 = self._resize(c3, size)
            if grad:
                return self.model(c3), o
            with torch.no_grad():
                return self.model(c3), o
        use_cache = self.cache_ori_cla is True or (self.cache_ori_cla is None and ori_key is not None
                                                   and ori_key[0] == 'ids' and self._classifier_is_frozen())
        if use_cache:
            mk = self._classifier_state_key()
            if mk != self._ori_cla_model_key:            # a weight or buffer was written (or replaced): every stored logit is stale
                self._ori_cla_cache.clear()
                self._ori_cla_model_key = mk
            if ori_key is None:
                o = ori_img() if callable(ori_img) else ori_img
                ori_key = (o.data_ptr(), o._version, tuple(o.shape))
                ori_img = o
            ori_cla = self._ori_cla_cache.get(ori_key)
            if ori_cla is not None:
                ori_cla = ori_cla.clone()            # the caller owns what it gets (the reference's callers write into logits in place, deepfool.py:54-57)
            else:
                if len(self._ori_cla_cache) >= 64:
                    self._ori_cla_cache.clear()
                ori_cla, o = ori_logits(False)
                self._ori_cla_cache[ori_key] = ori_cla.clone()       # (a private copy: the returned tensor is the caller's)
                if ori_key[0] != 'ids':              # an address-based key: the address must not be recycled while the key lives
                    keep = getattr(self, '_ori_keep_src', None)
                    self._ori_cla_keep = getattr(self, '_ori_cla_keep', [])[-63:] + [o if keep is None else keep]
        else:
            ori_cla, _ = ori_logits(True)
        return cla, ori_cla

    def _classifier_is_frozen(self):
        """The automatic logit cache's precondition: the classifier is a pure function of its input and nothing is being
        learned through it - every module in eval() mode (no dropout, no batch-norm statistics update) and no parameter
        requiring grad (the reference's attack loops set exactly this up: AS:281-287, attack_NeRFail.py:315-321)."""
        m = self.model
        if not isinstance(m, nn.Module):
            return False
        return not any(x.training for x in m.modules()) and not any(p.requires_grad for p in m.parameters())

    def _classifier_state_key(self):
        m = self.model
        if not isinstance(m, nn.Module):
            return None
        return tuple((t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))

    def _ori_key(self, views):
        """Cache key of the original images' logits: the views' ids (+ how often each id's resident image was replaced), else
        the identity of the device tensor the caller passed, else None."""
        vids = views.view_ids
        if vids is not None:
            return ('ids', tuple((k_, _VIEW_ORI_EPOCH.get(k_, 0)) for k_ in (_view_key(v, views.Ns) for v in vids)))
        return views.ori_src

    def attack_forward(self, spatial_rgb, weight_and_index_list, ori_img, view_ids=None):
        """The forward of one NeRFail-S step (AS:317) without what that step never uses: no `x` tensor (GN:83), no float copy
        of the images unless the original logits have to be computed, alpha + pass mask kept for the rgb-only backward.
        Returns (x_rgba leaf with requires_grad, cla, ori_cla, views, aux); differentiate cla down to x_rgba with autograd,
        then hot_backward_rgb(aux, x_rgba.grad, views)."""
        dev = _cuda()
        s = spatial_rgb.detach()
        if s.device != dev:
            s = s.to(dev)
        views = resolve_views(s, weight_and_index_list, ori_img, view_ids, self.keep_views_resident)
        _, x_rgba, aux = hot_forward(s, views, self.epsilon, self._mm() if self.update_epsilon_3d else None, need_x=False, need_aux=True)
        x_rgba.requires_grad_(True)
        self._ori_keep_src = getattr(views, '_ori_keep', None)
        cla, ori_cla = self.cold_tail(x_rgba, views.ori_float, ori_key=self._ori_key(views))
        self._ori_keep_src = None
        return x_rgba, cla, ori_cla, views, aux

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
        views = self._last_views
        B, P = views.B, views.P
        n = spatial_rgb.numel() // 4
        vi = views.indices()[0]                                                 # one view (batch 1): its per-view index
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


class _Compose(torch.autograd.Function):
    """x_rgba = compose(ori, r) of gauss_get_img (GN:309-319) on nerfail_gauss_compose; the gradient w.r.t. r is the plain chain
    rule of those three lines (elementwise, cold: nobody differentiates this in the reference's scripts)."""

    @staticmethod
    def forward(ctx, ori, r):
        out = torch.empty_like(r)
        _lib.check(_lib.load().nerfail_gauss_compose(_lib.dev(ori, 'ori_img'), _lib.dev(r, 'r'), r.numel() // 4, _lib.dev(out), _lib.stream()))
        ctx.save_for_backward(ori, r)
        return out

    @staticmethod
    def backward(ctx, g):
        ori, r = ctx.saved_tensors
        opaque = (ori[..., 3:4] > 0).to(g.dtype)
        g_rgb = g[..., :3] * opaque
        g_r = torch.cat([g_rgb * (r[..., 3:4] / 255), (g_rgb * r[..., :3]).sum(-1, keepdim=True) / 255], -1)
        return None, g_r


class gauss_get_r(nn.Module):
    """GN:189-268: the perturbation gathered onto the pixels of a batch of views straight from RAW distances,
    r = sum_k s[idx_k] g_k / (sum g + 0.001), g = exp(-(d/c)^2 / 2). Composition of K9 (nerfail_gauss_weight) and K10's gather
    (nerfail_gauss_fwd: its `x` output), differentiable w.r.t. spatial_rgb like the reference's."""

    def __init__(self, device, c, model, model_name):
        super(gauss_get_r, self).__init__()
        self.top_number = 8
        self.c = torch.nn.Parameter(torch.tensor([c]), requires_grad=False)
        self.device = device
        self.model = model
        self.model_name = model_name
        self.w = 299
        self.h = 299
        self.update_epsilon_3d = True
        self._eps_minmax = None

    _mm = gauss_net._mm
    epsilon_3d_max = gauss_net.epsilon_3d_max
    epsilon_3d_min = gauss_net.epsilon_3d_min
    epsilon_3d_zero = gauss_net.epsilon_3d_zero
    close_update_epsilon_3d = gauss_net.close_update_epsilon_3d
    open_update_epsilon_3d = gauss_net.open_update_epsilon_3d
    print_epsilon = gauss_net.print_epsilon

    def forward(self, spatial_rgb, dist_and_index_list):
        dev = _cuda()
        dai = _lib.f32c(dist_and_index_list, dev)
        if dai.dim() != 5 or dai.shape[1] != 2 or dai.shape[4] != 8:
            raise ValueError('dist_and_index_list must be [B,2,H,W,8] (CI:148-163)')
        if not float(self.c) > 0:
            raise _lib.NerfailError('gauss_get_r: c must be positive')
        wi = torch.ops.nerfail_mi.gauss_weight(dai, float(self.c))                         # K9
        ori0 = torch.zeros((dai.shape[0], dai.shape[2], dai.shape[3], 4), dtype=torch.float32, device=dev)   # r does not depend on the image
        x, _ = gauss_gather(spatial_rgb, wi, ori0, None, self._mm() if self.update_epsilon_3d else None, True)
        return x


class gauss_get_img(nn.Module):
    """GN:271-337: composite an already gathered r onto the images and classify. Hot part = nerfail_gauss_compose (no epsilon
    clip, no [0,255] clip - GN:309-319); the tail (NHWC -> NCHW, white background, Resize to 299 / 224 - here ALSO for
    "my_model", GN:339-347 - and the two classifier passes) is stock PyTorch like gauss_net's."""

    def __init__(self, device, c, model, model_name):
        super(gauss_get_img, self).__init__()
        self.top_number = 8
        self.c = torch.nn.Parameter(torch.tensor([c]), requires_grad=False)
        self.device = device
        self.model = model
        self.model_name = model_name
        self.w = 299
        self.h = 299

    _resize = gauss_net._resize

    def compose(self, ori_img, r):
        """(ori_img float [B,H,W,4], x_rgba): the part before the classifier."""
        dev = _cuda()
        ori = _lib.f32c(torch.as_tensor(ori_img), dev)                                     # GN:290
        r_ = r if (isinstance(r, torch.Tensor) and r.is_cuda and r.dtype == torch.float32 and r.is_contiguous()) else _lib.f32c(torch.as_tensor(r), dev)
        if ori.dim() != 4 or ori.shape[-1] != 4 or tuple(r_.shape) != tuple(ori.shape):
            raise ValueError('ori_img and r must both be [B,H,W,4], got %s and %s' % (tuple(ori.shape), tuple(r_.shape)))
        return ori, _Compose.apply(ori, r_)

    def forward(self, ori_img, r):
        ori, x_rgba = self.compose(ori_img, r)
        size = 224 if self.model_name == "vit_b_16" else 299

        def three(t):
            c = t.transpose(2, 3).transpose(1, 2)
            return self._resize(torch.where(c[:, 3:4] > 0, c[:, :3], torch.full_like(c[:, :3], 255.)).contiguous(), size)
        cla = self.model(three(x_rgba))
        ori_cla = self.model(three(ori))
        return r, x_rgba, cla, ori, ori_cla


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
