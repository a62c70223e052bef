"""Multi-GPU decomposition of the path (one process per GPU, torch.distributed; nccl = RCCL on ROCm).

Rays are independent units: a view's pixel range is cut into contiguous per-rank ranges and rendered
with NO data-path collective (each rank writes / keeps its own slice). The attack step has one real
exchange: the sum of the per-shard perturbation gradients (attack.nerfail_s_step).
The range arithmetic is pure host logic: importable and testable without a GPU.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous near-equal split of n units: returns [lo, hi) for `rank`. The first n % world ranks get one more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError('bad rank/world: %r/%r' % (rank, world))
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_ranges(n, world):
    return [shard_range(n, r, world) for r in range(world)]


def world_and_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


_STAGE = {}                 # (numel, dtype) -> pinned host buffer, insertion order = LRU order
STAGE_SLOTS = 4             # gradient sizes in use at once: one per perturbation table shape


def force_collectives():
    """NERFAIL_FORCE_COLLECTIVE=1: issue the all-reduce even in a 1-rank group (a dry run of the RCCL path on a one-GPU box:
    communicator creation, stream semantics and event timing are exercised; the sum over one rank is the identity)."""
    import os
    return os.environ.get('NERFAIL_FORCE_COLLECTIVE', '0') == '1'


def all_reduce_sum_(t, group=None):
    """In-place sum over ranks. RCCL all-reduce for HIP tensors whenever the group has a device-capable backend ('nccl', a
    'cpu:gloo,cuda:nccl' pair, or the default-initialised group that maps to it): the GPU-box path, xGMI. Only under a pure
    'gloo' group (the CPU tests and the one-GPU rehearsal, where RCCL refuses two ranks on one device) a HIP tensor is
    staged through a pinned host buffer (kept per size), reduced there and copied back."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    world = dist.get_world_size(group)
    if world <= 1 and not force_collectives():
        return t
    if t.is_cuda and str(dist.get_backend(group)) == 'gloo':
        key = (t.numel(), t.dtype)
        h = _STAGE.pop(key, None)
        if h is None:
            h = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
            while len(_STAGE) >= STAGE_SLOTS:                 # bounded: the least recently used size goes
                _STAGE.pop(next(iter(_STAGE)))
        _STAGE[key] = h                                       # most recently used last
        h.copy_(t.reshape(-1), non_blocking=False)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h.reshape(t.shape))
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def render_view_sharded(render_fn, H, W, rank, world):
    """Render this rank's pixel range of an H x W view. `render_fn(pix_begin, pix_count)` returns a dict of
    per-ray tensors; no collective is issued (the driver gathers or writes rank-local slices)."""
    lo, hi = shard_range(H * W, rank, world)
    return (lo, hi), render_fn(lo, hi - lo)


def render_shard(H, W, K, c2w, near, far, rank=None, world=None, chunk=1024 * 32, group=None, **render_kwargs):
    """This rank's contiguous pixel range of one view through the NC render pipeline (ray_gen of the range ->
    batchify_rays incl. pts_max). Returns ((lo, hi), dict of per-ray tensors [hi-lo, ...]). Rays are independent, so
    the per-rank slices concatenate BITWISE to the unsharded render (tests/test_hip_multigpu.py); no collective."""
    from . import run_nerf as RN
    if world is None:
        world, rank = world_and_rank(group)
    kw = {k: v for k, v in render_kwargs.items() if k not in ('ndc', 'use_viewdirs')}

    def render_fn(lo, n):
        if n == 0:
            return {}
        rays = RN.ray_gen(H, W, K, c2w, near, far, lo, n)
        return RN.batchify_rays(rays, chunk, want_pts_max=True, **kw)
    return render_view_sharded(render_fn, H, W, rank, world)
