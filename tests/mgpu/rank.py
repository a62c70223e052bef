#!/usr/bin/env python3
"""One rank of the multi-process PRODUCT-path test (tests/test_hip_multigpu.py). Started as a fresh process by
tests/mgpu/launcher.py with RANK / WORLD_SIZE / MASTER_* in the environment; every rank uses HIP device 0 (the test
box has one GPU), so the process group is gloo and sharding.all_reduce_sum_ stages the HIP gradient through pinned
host memory. What runs is nerfail_amd itself: attack.nerfail_s_step / sharded_perturbation_grad (AS:304-392 split
over ranks + the C1 all-reduce) and sharding.render_shard (ray range of one view, no collective).

    python tests/mgpu/rank.py OUTDIR          -> OUTDIR/w{world}_r{rank}.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path[:0] = [os.path.dirname(TESTS), TESTS]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth  # noqa: E402
from mgpu import problem as PB  # noqa: E402


def main(out_dir):
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from nerfail_amd import attack, sharding, nerf_to_coord as NC
    from nerfail_amd.GaussNet import gauss_net, create_gauss_w
    from nerfail_amd.run_nerf_helpers import NeRF
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {}

    # ---- attack: ITERS sequential NeRFail-S iterations on one ragged batch
    a = PB.attack_inputs()
    wi, _ = create_gauss_w(dev, 0.02)(T(a['dist_and_index']))
    cls_w = T(a['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ cls_w.t()
    net = gauss_net(dev, 0.02, Cls(), 'my_model', epsilon=None)
    s0, ori, label = T(a['s0']), T(a['ori']), torch.tensor(PB.LABEL, device=dev)
    g, loss = attack.sharded_perturbation_grad(net, s0, wi, ori, label)
    out['grad0'], out['loss0'] = g.cpu().numpy(), float(loss)
    s = s0
    for it in range(PB.ITERS):
        s, loss = attack.nerfail_s_step(net, s, s0, wi, ori, label, PB.A, PB.EPS, False)
        out['s_it%d' % it] = s.cpu().numpy()
    out['wi'] = wi.cpu().numpy()

    # ---- more ranks than views (attack.py, the `hi == lo` branches of both sharded gradients): ONE view on `world` ranks -
    # at world 2 rank 1 owns nothing and only takes part in the sum (AS:72, :304: 300 views in batches of 8 leave a last
    # batch of 4, so at 8 GPUs four ranks take this branch)
    lo1, hi1 = sharding.shard_range(1, rank, world)
    g1, l1 = attack.sharded_perturbation_grad(net, s0, wi[:1].contiguous(), ori[:1].contiguous(), label)
    b1 = attack.sharded_perturbation_grad_rgb(net, s0, wi[:1].contiguous(), ori[:1].contiguous(), label)
    out['one_view_grad'], out['one_view_loss'], out['one_view_buf'] = g1.cpu().numpy(), float(l1), b1.cpu().numpy()
    out['one_view_owned'] = hi1 - lo1

    # ---- cfg5's loop shape: nerfail_s_loop over 3 batches x 4 iterations, views named by dataset id (per-view indices
    # are built once, on the rank that owns the view in its batch's split)
    lp = PB.loop_inputs()
    wi_l, _ = create_gauss_w(dev, 0.02)(T(lp['dist_and_index']))
    ori_l, s0_l = T(lp['ori']), T(lp['s0'])
    batches = [(wi_l[b:b + PB.LOOP_BATCH].contiguous(), ori_l[b:b + PB.LOOP_BATCH].contiguous(), list(range(b, b + PB.LOOP_BATCH)))
               for b in range(0, PB.LOOP_VIEWS, PB.LOOP_BATCH)]
    trace = []
    s_l = attack.nerfail_s_loop(net, s0_l, s0_l, batches, label, PB.LOOP_ITERS, PB.A, PB.EPS, False,
                                on_iter=lambda it, b, s_, loss_: trace.append(s_.cpu().numpy()))
    out['loop_s'] = s_l.cpu().numpy()
    out['loop_trace'] = np.stack(trace).astype(np.int8)          # iterates are multiples of A within +-EPS: exact in int8

    # ---- render: this rank's ray range of one view (D=8 W=256 coarse+fine, 64+128 samples), no collective
    r = PB.render_inputs()

    def mk(seed):
        sd = synth.nerf_state_dict(seed=seed)
        m = NeRF(8, 256, 63, 27, 5, [4], True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev)
    coarse, fine = mk(r['seed_coarse']), mk(r['seed_fine'])
    kw = dict(network_query_fn=None, perturb=0., N_importance=128, network_fine=fine, N_samples=64, network_fn=coarse,
              white_bkgd=True, raw_noise_std=0.)
    with torch.no_grad():
        (lo, hi), part = sharding.render_shard(PB.RH_, PB.RW_, r['K'], torch.from_numpy(r['c2w']), 2., 6.,
                                               chunk=512, **kw)
    out['render_lo'], out['render_hi'] = lo, hi
    for k in ('rgb_map', 'disp_map', 'acc_map', 'pts_max', 'rgb0', 'z_std'):
        out['render_' + k] = part[k].cpu().numpy()
    if world == 1:                     # the unsharded reference call of the same product API
        with torch.no_grad():
            rgb, disp, acc, pts_max, ex = NC.render(PB.RH_, PB.RW_, r['K'], chunk=PB.RH_ * PB.RW_, c2w=torch.from_numpy(r['c2w']),
                                                    near=2., far=6., use_viewdirs=True, ndc=False, **kw)
        out['full_rgb_map'], out['full_pts_max'] = rgb.reshape(-1, 3).cpu().numpy(), pts_max.reshape(-1, 3).cpu().numpy()
        out['full_disp_map'], out['full_acc_map'] = disp.reshape(-1).cpu().numpy(), acc.reshape(-1).cpu().numpy()
    np.savez(os.path.join(out_dir, 'w%d_r%d.npz' % (world, rank)), **out)
    import ctypes                      # which native library served this process (driver-independent evidence)
    loaded = [l.split()[-1] for l in open('/proc/self/maps') if 'libnerfail_hip' in l]
    assert loaded, 'libnerfail_hip.so is not mapped: the product path did not run'
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print('rank %d/%d ok' % (rank, world), flush=True)


if __name__ == '__main__':
    main(sys.argv[1])
