"""Host logic of the multi-GPU decomposition, incl. a world_size-2 gloo run of the PRODUCT's
attack.sharded_perturbation_grad (shard range, k/B weighting, the one all-reduce) with a torch-CPU stand-in for
gauss_net (the HIP kernels have no CPU path; tests/test_hip_multigpu.py runs the same thing on the GPU with them)."""
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


def _torch_gauss_net(cls_w):
    """Differentiable torch-CPU stand-in with gauss_net's call contract (GN:46-159), so that the PRODUCT's
    attack.sharded_perturbation_grad (shard range, k/B weighting, the C1 all-reduce) runs unchanged on CPU ranks."""
    def net(s, wi, ori):
        w, idx = wi[:, 0], wi[:, 1].long()
        x = (s.reshape(-1, 4)[idx] * w[..., None]).sum(3)
        rgb = torch.where(ori[..., 3:4] > 0, ori[..., :3] + x[..., :3] * (x[..., 3:4] / 255.), torch.zeros(()))
        x_rgba = torch.clip(torch.cat([rgb, ori[..., 3:4]], -1), 0., 255.)
        c = x_rgba.permute(0, 3, 1, 2)
        img = torch.where(c[:, 3:4] > 0, c[:, :3], torch.full_like(c[:, :3], 255.))
        cla = torch.nn.functional.adaptive_avg_pool2d(img, 4).reshape(c.shape[0], -1) @ cls_w.t()
        return x, x_rgba, cla, ori, None
    return net


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
    from oracle import gauss as OG
    from mgpu import problem as PB
    from nerfail_amd import attack
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    a = PB.attack_inputs()                                      # 5 views over 2 ranks: ragged split 3 + 2
    wi, _ = OG.create_gauss_w(a['dist_and_index'])
    net = _torch_gauss_net(torch.from_numpy(a['cls_w']))
    s0, ori = torch.from_numpy(a['s0']), torch.from_numpy(a['ori'])
    g, loss = attack.sharded_perturbation_grad(net, s0, torch.from_numpy(wi), ori, torch.tensor(PB.LABEL))   # product code + C1
    out = OG.igsm_step(a['s0'], g.numpy(), a['s0'], PB.A, PB.EPS, False)
    np.save(os.path.join(out_dir, 'rank%d.npy' % rank), out)
    np.save(os.path.join(out_dir, 'grad%d.npy' % rank), g.numpy())
    dist.destroy_process_group()
    if rank == 0:                                               # the same product function, no process group: 1 rank
        full, loss1 = attack.sharded_perturbation_grad(net, s0, torch.from_numpy(wi), ori, torch.tensor(PB.LABEL))
        assert abs(float(loss) - float(loss1)) <= 1e-6 * abs(float(loss1))
        np.save(os.path.join(out_dir, 'full_grad.npy'), full.numpy())
        np.save(os.path.join(out_dir, 'single.npy'), OG.igsm_step(a['s0'], full.numpy(), a['s0'], PB.A, PB.EPS, False))


def test_attack_step_all_reduce_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1)                               # every rank applies the identical step
    assert np.array_equal(np.load(tmp_path / 'grad0.npy'), np.load(tmp_path / 'grad1.npy'))
    full, summed = np.load(tmp_path / 'full_grad.npy'), np.load(tmp_path / 'grad0.npy')
    assert np.abs(full).max() > 1e-6                            # a real gradient, not a saturated softmax's denormals
    assert np.abs(full - summed).max() <= 1e-5 * np.abs(full).max()   # N-rank sum == 1-rank gradient (fp32 order)
    single = np.load(tmp_path / 'single.npy')
    assert (r0 != single).mean() < 1e-3                         # sign() can flip only where |grad| ~ rounding


class _StandInViews:
    def __init__(self, Ns):
        self.Ns = Ns


class _StandInAttackNet:
    """gauss_net's attack_forward contract (GaussNet.py: x_rgba leaf, cla, ori_cla, views, aux) on torch-CPU, so that the
    PRODUCT's attack.perturbation_grad_rgb / sharded_perturbation_grad_rgb run unchanged on CPU ranks; `aux` carries what
    the stand-in of hot_backward_rgb needs to chain x_rgba's gradient down to the perturbation."""
    def __init__(self, cls_w):
        self.full = _torch_gauss_net(cls_w)
        self.cls_w = cls_w

    def hot(self, s, wi, ori):
        return self.full(s, wi, ori)[1]

    def attack_forward(self, s, wi, ori, view_ids=None):
        x_rgba = self.hot(s.detach(), wi, ori).detach().requires_grad_(True)
        c = x_rgba.permute(0, 3, 1, 2)
        img = torch.where(c[:, 3:4] > 0, c[:, :3], torch.full_like(c[:, :3], 255.))
        cla = torch.nn.functional.adaptive_avg_pool2d(img, 4).reshape(c.shape[0], -1) @ self.cls_w.t()
        return x_rgba, cla, None, _StandInViews(s.numel() // 4), (self, s.detach(), wi, ori)


def _stand_in_hot_backward_rgb(aux, grad_x_rgba, views, out=None):
    net, s, wi, ori = aux
    s = s.clone().requires_grad_(True)
    net.hot(s, wi, ori).backward(grad_x_rgba)
    out[:3 * views.Ns] = s.grad.reshape(-1, 4)[:, :3].reshape(-1)
    return out


def _worker_idle(rank, world, port, out_dir):
    """More ranks than views (attack.py: the `hi == lo` branches): 5 views on 8 ranks - ranks 5..7 own no view and only take
    part in the sum. The reference's loop has 300 views in batches of 8 (AS:72, :304): its last batch holds 4 views, so at
    8 ranks four of them take exactly this branch."""
    sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
    torch.set_num_threads(1)
    from oracle import gauss as OG
    from mgpu import problem as PB
    from nerfail_amd import attack, sharding as SH
    import nerfail_amd.GaussNet as GNm
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    a = PB.attack_inputs()
    assert PB.B < world and SH.shard_range(PB.B, world - 1, world) == (PB.B, PB.B)      # the last rank really is idle
    wi_np, _ = OG.create_gauss_w(a['dist_and_index'])
    wi, s0, ori, label = torch.from_numpy(wi_np), torch.from_numpy(a['s0']), torch.from_numpy(a['ori']), torch.tensor(PB.LABEL)
    cls_w = torch.from_numpy(a['cls_w'])
    lo, hi = SH.shard_range(PB.B, rank, world)
    # (1) the four-channel form
    g, loss = attack.sharded_perturbation_grad(_torch_gauss_net(cls_w), s0, wi, ori, label)
    # (2) the rgb-only form the NeRFail-S step uses (3 Ns + 1 floats, the loss in the tail), the HIP pieces replaced by
    # torch stand-ins; the shard / idle-rank / all-reduce logic is the product's
    attack._cuda = lambda: torch.device('cpu')
    GNm.hot_backward_rgb = _stand_in_hot_backward_rgb
    buf = attack.sharded_perturbation_grad_rgb(_StandInAttackNet(cls_w), s0, wi, ori, label)
    np.savez(os.path.join(out_dir, 'idle%d.npz' % rank), g=g.numpy(), loss=float(loss), buf=buf.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:                                               # the same product functions without a process group: 1 rank
        g1, loss1 = attack.sharded_perturbation_grad(_torch_gauss_net(cls_w), s0, wi, ori, label)
        buf1 = attack.sharded_perturbation_grad_rgb(_StandInAttackNet(cls_w), s0, wi, ori, label)
        np.savez(os.path.join(out_dir, 'idle_single.npz'), g=g1.numpy(), loss=float(loss1), buf=buf1.numpy())


def test_more_ranks_than_views_gloo_world8(tmp_path):
    """`B < world`. Idle ranks contribute zeros, every rank ends bit-identical, the sum equals the 1-rank
    gradient - for sharded_perturbation_grad and for sharded_perturbation_grad_rgb (attack.py, both `else` branches)."""
    world = 8
    mp.spawn(_worker_idle, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ('idle%d.npz' % k)) for k in range(world)]
    one = np.load(tmp_path / 'idle_single.npz')
    assert [int(x['hi']) - int(x['lo']) for x in r] == [1, 1, 1, 1, 1, 0, 0, 0]
    for x in r[1:]:
        assert np.array_equal(x['g'], r[0]['g']) and np.array_equal(x['buf'], r[0]['buf']) and float(x['loss']) == float(r[0]['loss'])
    scale = np.abs(one['g']).max()
    assert scale > 1e-6 and np.abs(r[0]['g'] - one['g']).max() <= 1e-5 * scale
    assert abs(float(r[0]['loss']) - float(one['loss'])) <= 1e-6 * abs(float(one['loss']))
    n3 = one['buf'].size - 1
    assert np.abs(r[0]['buf'][:n3] - one['buf'][:n3]).max() <= 1e-5 * np.abs(one['buf'][:n3]).max()
    assert abs(r[0]['buf'][n3] - one['buf'][n3]) <= 1e-6 * abs(one['buf'][n3])          # the loss travels in the tail
    # the two forms agree on the three channels the sign step reads
    assert np.abs(r[0]['buf'][:n3].reshape(-1, 3) - r[0]['g'].reshape(-1, 4)[:, :3]).max() <= 1e-5 * scale


def test_all_reduce_stages_through_host_only_off_nccl():
    """Without a process group the reduce is the identity (and does not touch the tensor)."""
    t = torch.arange(6.)
    assert sharding.all_reduce_sum_(t) is t and torch.equal(t, torch.arange(6.))
    assert sharding.world_and_rank() == (1, 0)


def test_render_shards_need_no_collective():
    calls = []

    def render_fn(lo, n):
        calls.append((lo, n))
        return {'rgb_map': torch.arange(lo, lo + n)}
    parts = [sharding.render_view_sharded(render_fn, 10, 10, r, 3) for r in range(3)]
    cat = torch.cat([p[1]['rgb_map'] for p in parts])
    assert torch.equal(cat, torch.arange(100))
    assert [p[0] for p in parts] == sharding.shard_ranges(100, 3)
