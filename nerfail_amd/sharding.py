"""Multi-GPU decomposition of the path (one process per GPU, torch.distributed; nccl = RCCL on ROCm).

Rays are independent units: a view's pixel range is cut into contiguous per-rank ranges and rendered
with NO data-path collective (each rank writes / keeps its own slice). The attack step has one real
exchange: the sum of the per-shard perturbation gradients (attack.nerfail_s_step).
Pure host logic: importable and testable without a GPU.
"""
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


def all_reduce_sum_(t, group=None):
    """In-place sum over ranks (RCCL all-reduce for HIP tensors, gloo for CPU tensors in the tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def render_view_sharded(render_fn, H, W, rank, world):
    """Render this rank's pixel range of an H x W view. `render_fn(pix_begin, pix_count)` returns a dict of
    per-ray tensors; no collective is issued (the driver gathers or writes rank-local slices)."""
    lo, hi = shard_range(H * W, rank, world)
    return (lo, hi), render_fn(lo, hi - lo)
