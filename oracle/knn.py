"""ORACLE (test infrastructure, not product code): exact 8-nearest-neighbour index build.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference: Create_spatial_point_set/create_index_and_dist.py:126-145 (CI). The reference ranks
with torch.cdist, which at these sizes takes the matmul path (|a|^2+|b|^2-2ab) and is
numerically noisy (0.18 % of its own ordered indices differ from exact
fp64 ordering; self-distances come out as ~5e-4 instead of 0), and with an unstable sort.
"Bit-exact" is therefore DEFINED here, and the HIP kernel must reproduce it bit for bit:

    d2(q,p) = ((qx-px)*(qx-px) + (qy-py)*(qy-py)) + (qz-pz)*(qz-pz)      float32, no FMA
    key     = (d2, global point index)   lexicographic, ascending
    output  = the 8 smallest keys; dist = sqrt(d2) correctly rounded float32

Pinned against fixture g8: (i) the index sets equal the exact fp64 sets wherever the fp64 gap
between the 8th and 9th neighbour exceeds fp32 resolution, and (ii) they agree with the
reference procedure's own output (torch.cdist + sort + merge, re-issued by make_golden.py)
wherever that output is itself well separated. Both rules are written out in
tests/test_oracle_knn.py.
"""
import numpy as np

F32 = np.float32


def d2_f32(q, p):
    """Squared distance with the defined float32 operation order. q [...,3], p [...,3] broadcastable."""
    dx = (q[..., 0] - p[..., 0]).astype(F32)
    dy = (q[..., 1] - p[..., 1]).astype(F32)
    dz = (q[..., 2] - p[..., 2]).astype(F32)
    return ((dx * dx + dy * dy).astype(F32) + dz * dz).astype(F32)


def knn8(queries, points, k=8, block=256):
    """queries [Nq,3] f32, points [M,3] f32 -> (dist [Nq,k] f32 ascending, idx [Nq,k] int32)."""
    q = np.ascontiguousarray(queries, F32).reshape(-1, 3)
    p = np.ascontiguousarray(points, F32).reshape(-1, 3)
    nq, m = q.shape[0], p.shape[0]
    out_d = np.empty((nq, k), F32)
    out_i = np.empty((nq, k), np.int32)
    ar = np.arange(m, dtype=np.int64)
    for s in range(0, nq, block):
        d2 = d2_f32(q[s:s + block, None, :], p[None, :, :])              # [b, M]
        # lexicographic (d2, index): stable argsort on d2 keeps ascending index among ties
        if m > 4 * k:
            part = np.argpartition(d2, 4 * k, axis=-1)[:, :4 * k + 1]
            kth = np.take_along_axis(d2, part, -1).max(-1, keepdims=True)
            # candidates = everything <= the (4k)-th value (keeps all ties), then exact ordering
            for r in range(d2.shape[0]):
                cand = ar[d2[r] <= kth[r, 0]]
                o = cand[np.argsort(d2[r, cand], kind='stable')][:k]
                out_i[s + r] = o
                out_d[s + r] = np.sqrt(d2[r, o])
        else:
            o = np.argsort(d2, axis=-1, kind='stable')[:, :k]
            out_i[s:s + block] = o
            out_d[s:s + block] = np.sqrt(np.take_along_axis(d2, o, -1))
    return out_d, out_i


def index_and_dist(view_pts, point_set):
    """File-level contract of CI:148-163: float32 [2,H,W,8] = (dist ascending, global index as float)."""
    H, W = view_pts.shape[:2]
    d, i = knn8(view_pts.reshape(-1, 3), point_set)
    return np.stack([d.reshape(H, W, 8), i.astype(F32).reshape(H, W, 8)], 0)
