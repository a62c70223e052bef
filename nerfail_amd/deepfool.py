"""MI355X mirror of deepfool.py (reference root, `deepfool(...)` lines 10-111): the per-view inner loop of NeRFail
(attack_NeRFail.py:396-402). Same signature and 5-tuple return (rot, loop_i, ori_cla_max_index, cla_max_index,
spatial_rgb).

The control flow (margins m1 / m2, per-class minimal step, accumulation, clamp, alpha restore) is host logic and
stays Python, as SURVEY.md section 2 #10 scopes it; every forward / backward through the pixel<->3-D map runs in the
HIP kernels behind `net` (gauss_net: K10 forward, K11 backward). Two savings over the reference that cannot change
the result: the gradient of the original class logit is computed once per iteration instead of once per competing
class (the reference recomputes the identical tensor up to 7 times at deepfool.py:76-77), and when `net` is
nerfail_amd's gauss_net all class gradients of an iteration come from ONE multi-right-hand-side pass over the
inverted index (gauss_net.logit_gradients -> nerfail_gauss_bwd_csr_multi) instead of one autograd.grad call each; the
competing class is then chosen on the device (first minimum, NaN never chosen: deepfool.py:85's strict `<`).
"""
import torch

from . import _lib


def _grad(out_scalar, wrt):
    return torch.autograd.grad(out_scalar, wrt, retain_graph=True, create_graph=False)[0]


def deepfool(net_input, e, net, num_classes=8, max_iter=20, target_label: int = None, overshoot: float = 0.02,
             m1: float = 1, m2: float = 30, universal_2d=False):
    if universal_2d:
        raise NotImplementedError('universal_2d is the 2-D baseline (attack_UAP_2D.py): out of scope, SURVEY.md section 2 #12')
    spatial_rgb, weight_and_index, ori_img = net_input
    if torch.cuda.is_available():          # the reference keeps everything on `device` (AN:354-362); CPU tensors are uploaded once
        dev = torch.device('cuda', torch.cuda.current_device())
        spatial_rgb = _lib.f32c(torch.as_tensor(spatial_rgb), dev)
        weight_and_index = _lib.f32c(torch.as_tensor(weight_and_index), dev)
        ori_img = _lib.f32c(torch.as_tensor(ori_img), dev)
    spatial_rgb_0 = spatial_rgb.clone().detach()
    spatial_rgb = spatial_rgb.detach().clone().requires_grad_(True)

    _, _, cla, _, ori_cla = net(spatial_rgb, weight_and_index, ori_img)                 # deepfool.py:33
    cla_max_index = torch.max(cla, 1)[1]
    ori_cla_max_index = torch.max(ori_cla, 1)[1]
    rot = torch.zeros_like(spatial_rgb_0)

    loop_i = 0
    while loop_i < max_iter:
        spatial_rgb = spatial_rgb.detach().clone().requires_grad_(True)
        x_out, x_rgba, cla, _, ori_cla = net(spatial_rgb, weight_and_index, ori_img)     # deepfool.py:51
        ori_cla_max_index = torch.max(ori_cla, 1)[1]
        o = int(ori_cla_max_index)
        bump = torch.zeros_like(cla)                                                    # deepfool.py:53-57 (+m1), out of place
        if target_label is None:
            bump[:, o] = m1
        else:
            bump[:, :int(target_label)] = m1
            bump[:, int(target_label) + 1:] = m1
        cla = cla + bump
        cla_max_index = torch.max(cla, 1)[1]
        if (target_label is None) and int(cla_max_index) != o:
            break
        if (target_label is not None) and int(cla_max_index) == int(target_label):
            break

        if target_label is None:
            ks = [k for k in range(num_classes) if k != o]
        else:
            ks = [int(target_label)]
        # the multi-RHS pass takes 2..8 logits ([o] + ks): with o >= num_classes all num_classes competitors remain (9 at
        # num_classes = 8), with a single class none does - those cases iterate class by class like the reference
        multi = hasattr(net, 'logit_gradients') and getattr(net, 'deterministic', False) and 2 <= len(ks) + 1 <= 8
        f_prime = (cla[0, ks] - (cla[0, o] + m2)).detach()                              # deepfool.py:79
        if multi:
            # all class gradients in one pass over the inverted index (K11 multi-RHS), then the step arithmetic in two
            # streaming kernels (K14): ||G_k - G_o||^2 for every k, device-side choice, rot / clamp / alpha restore
            G = net.logit_gradients(spatial_rgb, None, x_out, x_rgba, cla, [o] + ks)
            lib = _lib.load()
            C, n = G.shape[0], G[0].numel() // 4
            nb = lib.nerfail_deepfool_norms_scratch_bytes(C, n)
            scratch = torch.empty((nb,), dtype=torch.uint8, device=G.device)
            norms2 = torch.empty((C - 1,), dtype=torch.float32, device=G.device)
            _lib.check(lib.nerfail_deepfool_norms(_lib.dev(G), C, n, _lib.dev(scratch), nb, _lib.dev(norms2), _lib.stream()))
            nrm = torch.sqrt(norms2)                                                    # torch.norm(grad_prime), every k
        else:
            grad_o = _grad(cla[:, o].sum(), spatial_rgb)
            grads_k = torch.stack([_grad(cla[:, k].sum(), spatial_rgb) for k in ks])
            grad_prime = grads_k - grad_o.unsqueeze(0)                                  # deepfool.py:78 for every k
            nrm = torch.linalg.vector_norm(grad_prime.reshape(len(ks), -1), dim=1)
        if target_label is None:
            value_r = torch.abs(f_prime) / (nrm + 0.0001)                               # deepfool.py:81
            value_r = torch.where(torch.isnan(value_r), torch.full_like(value_r, float('inf')), value_r)
            best = torch.argmin(value_r)                                                # first minimum = strict `<` scan
            scale = torch.abs(f_prime[best]) / ((nrm[best] ** 2) + 0.0001)              # deepfool.py:86
            scale = torch.where(torch.isinf(value_r[best]), torch.zeros_like(scale), scale)   # nothing chosen: dr = 0
        else:
            best = torch.zeros((), dtype=torch.int64, device=nrm.device)
            scale = torch.abs(f_prime[0]) / ((nrm[0] ** 2) + 0.0001)                    # deepfool.py:92-96
        if multi:
            rot = _lib.f32c(rot)
            new_s = torch.empty_like(spatial_rgb_0)
            # the device scalars are bound to names: a temporary freed right after its pointer was taken could be handed
            # to the next allocation and overwritten before the kernel (enqueued afterwards) reads it
            best_i, scale_f, s0_c = (best + 1).to(torch.int32), scale.to(torch.float32).reshape(1), _lib.f32c(spatial_rgb_0)
            _lib.check(lib.nerfail_deepfool_apply(_lib.dev(G), C, n, _lib.dev(best_i), _lib.dev(scale_f), float(overshoot),
                                                  _lib.dev(s0_c), _lib.dev(rot), _lib.dev(new_s), _lib.stream()))
            spatial_rgb = new_s                                                         # deepfool.py:98-102 fused
        else:
            rot = (rot + scale * grad_prime[best]).detach()
            spatial_rgb = torch.clamp((spatial_rgb_0 + (overshoot * rot)).detach(), -255, 255)
            spatial_rgb = torch.cat([spatial_rgb[:, :, :, :3], spatial_rgb_0[:, :, :, 3].unsqueeze(-1)], -1)   # alpha unchanged
        loop_i += 1

    spatial_rgb = spatial_rgb.detach()
    rot = (spatial_rgb - spatial_rgb_0).detach()
    return rot, loop_i, ori_cla_max_index, cla_max_index, spatial_rgb
