"""CPU restatement (numpy, float32) of one torch.optim.Adam step as the reference's training loop runs it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). The reference builds `torch.optim.Adam(params, lr=lrate,
betas=(0.9, 0.999))` (run_nerf.py:207), calls `optimizer.step()` (RN:792) and rewrites `param_group['lr']` with the
exponential decay of RN:796-800. The arithmetic lives in PyTorch (third-party, torch/optim/adam.py
`_single_tensor_adam` on CPU tensors); it is restated here operation by operation and pinned to fixture
tests/golden/g13_adam.npz, produced by running that optimizer (tests/golden/make_golden.py g13): m and v bit-exact,
parameters bit-exact except ~1 element in 1000 that differs by one ulp.
"""
import numpy as np

F = np.float32


def _fma(a, b, c):
    # products of two float32 are exact in float64; the sum is rounded once more to float32 (double rounding is
    # possible in principle, observed on none of the fixture's elements)
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def bias_corrections(lr, step, beta1=0.9, beta2=0.999):
    """(step_size, bias_correction2_sqrt) in Python doubles, as torch/optim/adam.py computes them."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    return lr / bc1, bc2 ** 0.5


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """One step for one tensor; returns (p, m, v). `step` is the 1-based step count AFTER the increment."""
    p, g, m, v = (np.asarray(a, F) for a in (p, g, m, v))
    m = _fma(F(1 - beta1), g - m, m)                         # exp_avg.lerp_(grad, 1 - beta1): vectorised lerp is an fma
    v = _fma(F(1 - beta2) * g, g, v * F(beta2))              # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    step_size, bc2_sqrt = bias_corrections(lr, step, beta1, beta2)
    denom = np.sqrt(v) / F(bc2_sqrt) + F(eps)                # (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p + (F(-step_size) * m) / denom                      # param.addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(F), m, v


def decayed_lrate(lrate, global_step, lrate_decay, decay_rate=0.1):
    """RN:796-798."""
    return lrate * (decay_rate ** (global_step / (lrate_decay * 1000)))
