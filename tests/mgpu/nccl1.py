#!/usr/bin/env python3
"""RCCL dry run on ONE GPU (tests/test_hip_multigpu.py::test_rccl_path_one_rank_dry_run): a 1-rank
process group with backend 'nccl' (= RCCL on ROCm), bound to the device, and the PRODUCT's gradient all-reduce issued
through it (NERFAIL_FORCE_COLLECTIVE=1: sharding.all_reduce_sum_ does not short-cut the 1-rank group). What this executes
that the gloo tests cannot: init_process_group('nccl', device_id=...), communicator creation, dist.all_reduce on the real
23 MB-class HIP buffer on torch's current stream, HIP-event timing around it, MIN/MAX reductions of a checksum.

    python tests/mgpu/nccl1.py OUTDIR          -> OUTDIR/nccl1.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path[:0] = [os.path.dirname(TESTS), TESTS]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mgpu import problem as PB  # noqa: E402


def main(out_dir):
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    from nerfail_amd import attack
    from nerfail_amd.GaussNet import gauss_net, create_gauss_w
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    a = PB.attack_inputs()
    wi, _ = create_gauss_w(dev, 0.02)(T(a['dist_and_index']))
    cls_w = T(a['cls_w'])

    class Cls(torch.nn.Module):
        def forward(self, x):
            return torch.nn.functional.adaptive_avg_pool2d(x, 4).reshape(x.shape[0], -1) @ cls_w.t()
    net = gauss_net(dev, 0.02, Cls(), 'my_model', epsilon=None)
    s0, ori, label = T(a['s0']), T(a['ori']), torch.tensor(PB.LABEL, device=dev)
    ref = attack.sharded_perturbation_grad_rgb(net, s0, wi, ori, label).clone()          # no process group: no collective
    s_ref, _ = attack.nerfail_s_step(net, s0, s0, wi, ori, label, PB.A, PB.EPS, False)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    os.environ['NERFAIL_FORCE_COLLECTIVE'] = '1'
    timing = {}
    got = attack.sharded_perturbation_grad_rgb(net, s0, wi, ori, label, timing=timing)
    s_got, _ = attack.nerfail_s_step(net, s0, s0, wi, ori, label, PB.A, PB.EPS, False, timing=timing)
    torch.cuda.synchronize()
    ev = timing['allreduce_events']
    ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
    chk = got.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, 'nccl1.npz'), ref=ref.cpu().numpy(), got=got.cpu().numpy(), s_ref=s_ref.cpu().numpy(),
             s_got=s_got.cpu().numpy(), allreduce_ms=np.array(ms), allreduce_bytes=np.array([b for _, _, b in ev]),
             backend=np.array(str(dist.get_backend())), minmax=np.array([float(lo[0]), float(hi[0]), float(chk[0])]))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
