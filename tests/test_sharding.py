"""Host logic of the multi-GPU decomposition, incl. a world_size-2 gloo run of the attack step's one
collective (perturbation-gradient all-reduce). Compute inside the ranks is the ORACLE (the product has no CPU
path); what is under test is nerfail_amd.sharding / the reduce-then-step structure of attack.nerfail_s_step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfail_amd import sharding


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 640000, 640001):
        for world in (1, 2, 3, 8):
            rs = sharding.shard_ranges(n, world)
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
    from oracle import gauss as OG
    import synth
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(0)
    P, B, H, W = 3, 5, 12, 12                                   # 5 views over 2 ranks: ragged split 3 + 2
    s = rs.uniform(-20, 20, (P, H, W, 4)).astype(np.float32)
    s[..., 3] = 255.0
    ori = synth.disc_alpha_image(B, H, W, seed=1)
    dist_ = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, (B, H, W, 8)).astype(np.float32)
    wi, _ = OG.create_gauss_w(np.stack([dist_, idx], 1))
    G = rs.normal(size=(B, H, W, 4)).astype(np.float32) / B     # stands in for d(mean CE)/d(x_rgba)
    lo, hi = sharding.shard_range(B, rank, world)
    g = OG.gauss_backward(s, wi[lo:hi], ori[lo:hi], np.zeros_like(G[lo:hi]), G[lo:hi], None)
    gt = torch.from_numpy(g)
    sharding.all_reduce_sum_(gt)                                # C1
    out = OG.igsm_step(s, gt.numpy(), s, 2.0, 32.0, False)
    np.save(os.path.join(out_dir, 'rank%d.npy' % rank), out)
    if rank == 0:
        full = OG.gauss_backward(s, wi, ori, np.zeros_like(G), G, None)
        np.save(os.path.join(out_dir, 'full_grad.npy'), full)
        np.save(os.path.join(out_dir, 'sum_grad.npy'), gt.numpy())
        np.save(os.path.join(out_dir, 'single.npy'), OG.igsm_step(s, full, s, 2.0, 32.0, False))
    dist.destroy_process_group()


def test_attack_step_all_reduce_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1)                               # every rank applies the identical step
    full, summed = np.load(tmp_path / 'full_grad.npy'), np.load(tmp_path / 'sum_grad.npy')
    assert np.abs(full - summed).max() <= 1e-5 * np.abs(full).max()   # N-rank sum == 1-rank gradient (fp32 order)
    single = np.load(tmp_path / 'single.npy')
    assert (r0 != single).mean() < 1e-3                         # sign() can flip only where |grad| ~ rounding


def test_render_shards_need_no_collective():
    calls = []

    def render_fn(lo, n):
        calls.append((lo, n))
        return {'rgb_map': torch.arange(lo, lo + n)}
    parts = [sharding.render_view_sharded(render_fn, 10, 10, r, 3) for r in range(3)]
    cat = torch.cat([p[1]['rgb_map'] for p in parts])
    assert torch.equal(cat, torch.arange(100))
    assert [p[0] for p in parts] == sharding.shard_ranges(100, 3)
