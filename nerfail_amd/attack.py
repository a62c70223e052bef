"""The attack-step arithmetic of attack_NeRFail_S.py (reference = AS), on the MI355X.

`igsm_step` is AS:352-392 as one kernel. `nerfail_s_step` is one iteration of the AS:304-392 loop body
(forward through gauss_net, CE loss, backward to the perturbation, sign step, epsilon clamp); with
torch.distributed initialised it shards the batch's views over ranks and sums the perturbation gradient
with ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) before every rank applies the
identical step - the only collective on the whole path.
"""
import torch
import torch.distributed as dist

from . import _lib
from . import ops  # noqa: F401  (registers torch.ops.nerfail_mi.*)
from . import sharding
from .run_nerf_helpers import _cuda


def igsm_step(spatial, grad, spatial_init, a=2.0, epsilon=32.0, targeted=False, out=None):
    """AS:352-392: rgb <- rgb -/+ a*sign(grad) where alpha > 0 else 0; clamp to init +- epsilon; alpha kept."""
    dev = _cuda()
    s = _lib.f32c(spatial, dev)
    g = _lib.f32c(grad, dev)
    s0 = _lib.f32c(spatial_init, dev)
    if s.shape[-1] != 4 or g.shape != s.shape or s0.shape != s.shape:
        raise ValueError('spatial, grad and spatial_init must all be [..., 4] of the same shape')
    res = torch.ops.nerfail_mi.igsm_step(s, g, s0, float(a), float(epsilon), bool(targeted))     # K12 as a registered op
    if out is not None:
        out.copy_(res)
        return out
    return res


def perturbation_grad(net, spatial, weight_and_index, ori_img, label, batch_total=None, grad_fn=None, view_ids=None):
    """d(CE(cla, label))/d(spatial) for the views given (AS:317-348). `batch_total` = views in the WHOLE batch
    (CE is a mean over the batch, so a shard holding k of B views contributes with weight k/B). `view_ids`: the views'
    dataset indices - keys of their cached / persisted inverted indices (GaussNet.view_indices, load_view_indices)."""
    s = spatial.detach().clone().requires_grad_(True)
    if view_ids is not None:
        x, r, cla, ori, ori_cla = net(s, weight_and_index, ori_img, view_ids=view_ids)
    else:
        x, r, cla, ori, ori_cla = net(s, weight_and_index, ori_img)
    lab = label.to(cla.device).broadcast_to([cla.shape[0]])
    if grad_fn is not None:
        loss = grad_fn(cla, lab)
    else:
        loss = torch.nn.functional.cross_entropy(cla, lab, reduction='sum') / float(batch_total or cla.shape[0])
    loss.backward()
    return s.grad, loss.detach(), cla.detach()


def igsm_step_rgb(spatial, grad_rgb, spatial_init, a=2.0, epsilon=32.0, targeted=False):
    """AS:352-392 with the gradient as [Ns,3] (rgb only; the alpha channel's gradient is never read by the sign step)."""
    dev = _cuda()
    s, s0 = _lib.f32c(spatial, dev), _lib.f32c(spatial_init, dev)
    n = s.numel() // 4
    g = _lib.f32c(grad_rgb, dev)
    if s.shape[-1] != 4 or s0.shape != s.shape or g.numel() < 3 * n:
        raise ValueError('spatial / spatial_init must be [..., 4] of the same shape and grad_rgb hold 3 floats per row')
    out = torch.empty_like(s)
    _lib.check(_lib.load().nerfail_igsm_step_rgb(_lib.dev(s), _lib.dev(g), _lib.dev(s0), n, float(a), float(epsilon), int(bool(targeted)),
                                                 _lib.dev(out), _lib.stream()))
    return out


def perturbation_grad_rgb(net, spatial, weight_and_index, ori_img, label, batch_total=None, view_ids=None, out=None):
    """The NeRFail-S step's gradient in the form the step consumes (AS:357-392 reads grad[..., :3] only): a flat buffer of
    3 Ns + 1 floats - d(CE)/d(spatial rgb) as [Ns,3], then the loss - filled by the rgb-only backward (no `x` tensor, 5 bytes
    per pixel between forward and backward, 23 MB instead of 30.7 MB for the all-reduce, which carries the loss in the same
    collective). The three channels equal perturbation_grad()'s bit for bit."""
    from .GaussNet import hot_backward_rgb
    xr, cla, ori_cla, views, aux = net.attack_forward(spatial, weight_and_index, ori_img, view_ids)
    lab = label.to(cla.device).broadcast_to([cla.shape[0]])
    loss = torch.nn.functional.cross_entropy(cla, lab, reduction='sum') / float(batch_total or cla.shape[0])
    loss.backward()
    Ns = views.Ns
    if out is None:
        out = torch.empty((3 * Ns + 1,), dtype=torch.float32, device=xr.device)
    hot_backward_rgb(aux, xr.grad, views, out)
    out[3 * Ns] = loss.detach()
    return out, cla.detach()


def sharded_perturbation_grad_rgb(net, spatial, weight_and_index, ori_img, label, group=None, timing=None, view_ids=None):
    """perturbation_grad_rgb of this rank's share of the batch's views, then ONE all-reduce of the 3 Ns + 1 floats (C1: the
    gradient and, in its tail, the loss). Identical on every rank."""
    world, rank = sharding.world_and_rank(group)
    if view_ids is None and getattr(weight_and_index, 'view_ids', None) is not None:
        view_ids = weight_and_index.view_ids                # a MyDataset.collate_views batch: ids travel with the list
    B = len(view_ids) if view_ids is not None else weight_and_index.shape[0]
    lo, hi = sharding.shard_range(B, rank, world)
    Ns = spatial.numel() // 4
    if hi > lo:
        buf, _ = perturbation_grad_rgb(net, spatial, None if weight_and_index is None else weight_and_index[lo:hi],
                                       None if ori_img is None else ori_img[lo:hi], label, batch_total=B,
                                       view_ids=None if view_ids is None else list(view_ids)[lo:hi])
    else:                                   # more ranks than views: this rank only takes part in the sum
        buf = torch.zeros((3 * Ns + 1,), dtype=torch.float32, device=_cuda())
    if world > 1 or sharding.force_collectives():
        if timing is not None and buf.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        sharding.all_reduce_sum_(buf, group)         # C1: the perturbation-gradient all-reduce (+ the loss in its tail)
        if timing is not None and buf.is_cuda:
            e1.record()
            timing.setdefault('allreduce_events', []).append((e0, e1, buf.numel() * buf.element_size()))
    return buf


def sharded_perturbation_grad(net, spatial, weight_and_index, ori_img, label, group=None, timing=None, view_ids=None):
    """d(mean CE over the WHOLE batch)/d(spatial), identical on every rank: this rank differentiates its contiguous
    share of the batch's views (weight k/B), then ONE all-reduce sums the [P,H,W,4] gradient (C1).
    `timing`: optional dict; gets HIP events around the collective ('allreduce_events') for bench.py."""
    world, rank = sharding.world_and_rank(group)
    B = weight_and_index.shape[0]
    lo, hi = sharding.shard_range(B, rank, world)
    if hi > lo:
        g, loss, _ = perturbation_grad(net, spatial, weight_and_index[lo:hi], ori_img[lo:hi], label, batch_total=B,
                                       view_ids=None if view_ids is None else list(view_ids)[lo:hi])
    else:                                   # more ranks than views: this rank only takes part in the sum
        g = torch.zeros_like(spatial)
        loss = torch.zeros((), device=spatial.device)
    if world > 1:
        if timing is not None and g.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        sharding.all_reduce_sum_(g, group)           # C1: the perturbation-gradient all-reduce
        if timing is not None and g.is_cuda:
            e1.record()
            timing.setdefault('allreduce_events', []).append((e0, e1, g.numel() * g.element_size()))
        sharding.all_reduce_sum_(loss, group)
    return g, loss


def perturbation_step_rgb(net, spatial, spatial_init, weight_and_index, ori_img, label, a, epsilon, targeted, view_ids=None):
    """One rank, no collective: forward, CE, classifier backward, then the gather backward with the sign step AS:352-392 as its
    epilogue (GaussNet.hot_backward_rgb_step) - the [Ns,3] gradient is never materialised. Returns (new perturbation, loss);
    bit-identical to perturbation_grad_rgb + igsm_step_rgb."""
    from .GaussNet import hot_backward_rgb_step
    xr, cla, ori_cla, views, aux = net.attack_forward(spatial, weight_and_index, ori_img, view_ids)
    lab = label.to(cla.device).broadcast_to([cla.shape[0]])
    loss = torch.nn.functional.cross_entropy(cla, lab, reduction='sum') / float(cla.shape[0])
    loss.backward()
    return hot_backward_rgb_step(aux, xr.grad, views, spatial, spatial_init, a, epsilon, targeted), loss.detach()


def nerfail_s_step(net, spatial, spatial_init, weight_and_index, ori_img, label, a=2.0, epsilon=32.0,
                   targeted=False, group=None, timing=None, view_ids=None):
    """One NeRFail-S iteration (AS:304-392) on one batch of views. Sharded over ranks when torch.distributed is up:
    every rank ends with the identical perturbation tensor."""
    if getattr(net, 'deterministic', True) and getattr(net, 'rgb_grad_only', True):
        world, _ = sharding.world_and_rank(group)
        if world == 1 and not sharding.force_collectives() and getattr(net, 'fused_sign_step', True):
            if view_ids is None and getattr(weight_and_index, 'view_ids', None) is not None:
                view_ids = weight_and_index.view_ids
            out, loss = perturbation_step_rgb(net, spatial, spatial_init, weight_and_index, ori_img, label, a, epsilon, targeted, view_ids)
            return out.view(spatial.shape), loss
        Ns = spatial.numel() // 4
        buf = sharded_perturbation_grad_rgb(net, spatial, weight_and_index, ori_img, label, group, timing, view_ids)
        return igsm_step_rgb(spatial, buf, spatial_init, a, epsilon, targeted), buf[3 * Ns]
    g, loss = sharded_perturbation_grad(net, spatial, weight_and_index, ori_img, label, group, timing, view_ids)
    return igsm_step(spatial, g, spatial_init, a, epsilon, targeted), loss


def nerfail_s_loop(net, spatial, spatial_init, batches, label, iters, a=2.0, epsilon=32.0, targeted=False, group=None,
                   on_iter=None):
    """The AS:278-392 loop shape of BASELINE configs[2]: `iters` passes over `batches` (list of (weight_and_index,
    ori_img[, view_ids]) per batch of views), the perturbation updated after EVERY batch (sequential dependence, AS:306-392).
    Returns the final perturbation; `on_iter(it, b, s, loss)` sees every iterate."""
    s = spatial
    for it in range(iters):
        for b, batch in enumerate(batches):
            wi, ori = batch[0], batch[1]
            s, loss = nerfail_s_step(net, s, spatial_init, wi, ori, label, a, epsilon, targeted, group,
                                     view_ids=batch[2] if len(batch) > 2 else None)
            if on_iter is not None:
                on_iter(it, b, s, loss)
    return s
