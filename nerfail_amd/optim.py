"""Optimizer of the NeRF training loop on the HIP path.

`Adam` is a drop-in for the reference's `torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))`
(run_nerf.py:207): same constructor, same `param_groups` (the lr-decay lines RN:796-800 keep working verbatim), same
`state_dict()` layout (step / exp_avg / exp_avg_sq per parameter, so `optimizer_state_dict` of a reference checkpoint
loads, RN:219). Only `step()` differs: all parameter tensors are updated by ONE kernel (nerfail_adam_step) instead of
torch's ~10 foreach launches.
"""
import torch

from . import _lib


def decayed_lrate(lrate, global_step, lrate_decay, decay_rate=0.1):
    """RN:796-798: new_lrate = lrate * decay_rate ** (global_step / (lrate_decay * 1000))."""
    return lrate * (decay_rate ** (global_step / (lrate_decay * 1000)))


class Adam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            if group['weight_decay'] != 0 or group['amsgrad'] or group.get('maximize', False):
                raise NotImplementedError('nerfail_amd.optim.Adam implements the reference configuration only '
                                          '(no weight decay / amsgrad / maximize)')
            beta1, beta2 = group['betas']
            lr = float(group['lr'])
            entries, keep, touched = [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('nerfail_amd.optim.Adam: parameters must be contiguous float32 tensors on the GPU')
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients')
                state = self.state[p]
                if len(state) == 0:                               # torch/optim/adam.py _init_group
                    state['step'] = torch.tensor(0.0, dtype=torch.float32)
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if state['step'].is_cuda:                         # a checkpoint saved from a capturable optimizer
                    state['step'] = state['step'].cpu()
                state['step'] += 1
                t = float(state['step'])
                g = _lib.f32c(p.grad)
                keep.append(g)
                touched.append(p)
                e = _lib.AdamTensor()
                e.param, e.grad = p.data_ptr(), g.data_ptr()
                e.exp_avg, e.exp_avg_sq = state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr()
                e.numel = p.numel()
                e.step_size = lr / (1 - beta1 ** t)
                e.bias_correction2_sqrt = (1 - beta2 ** t) ** 0.5
                entries.append(e)
            if entries:
                arr = (_lib.AdamTensor * len(entries))(*entries)
                _lib.check(lib.nerfail_adam_step(arr, len(entries), float(beta1), float(beta2), float(group['eps']),
                                                 _lib.stream()))
                # the kernel wrote through raw pointers: tell autograd (and the packed-weight caches keyed on
                # Tensor._version) that these tensors changed in place
                for p in touched:
                    torch.autograd.graph.increment_version(p)
                    torch.autograd.graph.increment_version(self.state[p]['exp_avg'])
                    torch.autograd.graph.increment_version(self.state[p]['exp_avg_sq'])
        return loss
