"""-m gpu: the error branches of the Python mirrors (engineering: "neither tested for line coverage of its error
branches"). The reference has no error conventions of its own (asserts only); the mirrors check what the kernels
index with - shapes, sizes, ids - BEFORE any launch, because a mismatched shape would read out of bounds. Every `raise` of
GaussNet.py / attack.py / run_nerf.py / run_nerf_helpers.py / optim.py / create_index_and_dist.py / deepfool.py that a caller can
reach with wrong arguments is reached here, and nothing is left half-registered behind a refused call."""
import numpy as np
import pytest
import torch

import synth
from hiputil import T, dev, hip_nerf

pytestmark = pytest.mark.gpu


def _toy(B=2, P=3, H=6, W=5, seed=3):
    from nerfail_amd import GaussNet as G
    rs = np.random.RandomState(seed)
    s = rs.uniform(-10, 10, (P, H, W, 4)).astype(np.float32)
    s[..., 3] = 255.0
    ori = synth.disc_alpha_image(B, H, W, seed=seed + 1)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, (B, H, W, 8)).astype(np.float32)
    wi, _ = G.create_gauss_w(dev(), 0.02)(T(np.stack([dist, idx], 1)))
    return T(s), wi, T(ori)


@pytest.fixture(autouse=True)
def _clean_caches():
    from nerfail_amd import GaussNet as G
    for c in (G._VIEW_CACHE, G._VIEW_MAPS, G._VIEW_ORI):
        c.clear()
    yield
    for c in (G._VIEW_CACHE, G._VIEW_MAPS, G._VIEW_ORI):
        c.clear()


def test_gauss_inputs_are_checked_before_any_launch():
    from nerfail_amd import GaussNet as G
    s, wi, ori = _toy()
    with pytest.raises(ValueError, match=r'\[\.\.\., 4\]'):
        G.gauss_gather(s[..., :3].contiguous(), wi, ori)                              # table rows must be BGRA
    with pytest.raises(ValueError, match='B,2,H,W,8'):
        G.gauss_gather(s, wi[:, :, :, :, :7].contiguous(), ori)                       # 8 neighbours per pixel
    with pytest.raises(ValueError, match='B,2,H,W,8'):
        G.gauss_gather(s, wi[:, 0], ori)                                              # weights AND indices
    with pytest.raises(ValueError, match='ori_img must be'):
        G.gauss_gather(s, wi, ori[:1])                                                # one image per view
    with pytest.raises(ValueError, match='ori_img must be'):
        G.gauss_gather(s, wi, ori[:, :-1].contiguous())                               # ... of the map's size
    with pytest.raises(ValueError, match='same length'):
        G.resolve_views(s, [wi[0], wi[1]], [ori[0]])                                  # per-view lists of different lengths
    with pytest.raises(ValueError, match=r'\[2,H,W,8\] map'):
        G.resolve_views(s, [wi[0], wi[1][:1]], [ori[0], ori[1]])
    x, xr = G.gauss_gather(s, wi, ori)                                                # ... and the good call still works
    assert torch.isfinite(xr).all()


def test_view_ids_are_checked():
    from nerfail_amd import GaussNet as G
    s, wi, ori = _toy()
    Ns = s.numel() // 4
    with pytest.raises(ValueError, match='name every view'):
        G.view_indices(wi, Ns, view_ids=[('e', 0)])                                   # 2 views, 1 id
    with pytest.raises(KeyError, match='neither a cached index nor a map'):
        G.view_indices([None], Ns, view_ids=[('e', 'unknown')])
    G.view_indices(wi, Ns, view_ids=[('e', 0), ('e', 1)])
    # the same id with another resolution / table size names another view: refused, and the good entry survives
    s2, wi2, ori2 = _toy(H=7, W=5)
    with pytest.raises(ValueError, match='names another view'):
        G.view_indices(wi2, Ns, view_ids=[('e', 0), ('e', 1)])
    assert len(G.view_indices(wi, Ns, view_ids=[('e', 0), ('e', 1)])) == 2
    # an index registered under an id that already names ANOTHER map is refused unless replace=True
    other = G.ViewIndex(wi[1], Ns)
    small = G.ViewIndex(wi2[0], s2.numel() // 4)
    with pytest.raises(ValueError, match='already names the index of another map'):
        G.register_view_index(('e', 0), other)
    with pytest.raises(ValueError, match='already names the index of another map'):
        G.register_view_index(('e', 0), small, Ns=Ns)
    G.register_view_index(('e', 0), other, replace=True)
    # ... and the first use of that id with the ORIGINAL map at hand notices the swap (fingerprint mismatch)
    other.verified = False
    with pytest.raises(ValueError, match='fingerprint mismatch'):
        G.view_indices(wi, Ns, view_ids=[('e', 0), ('e', 1)])
    # resident views: a map of the wrong shape, a batch that names a view nobody registered, a reused id with another map
    with pytest.raises(ValueError, match=r'\[2,H,W,8\]'):
        G.register_view(('r', 0), Ns, weight_and_index=wi[0][:, :, :, :4])
    G.register_view(('r', 0), Ns, weight_and_index=wi[0], ori_img=ori[0])
    with pytest.raises(KeyError, match='not resident'):
        G.resolve_views(s, None, None, view_ids=[('r', 0), ('r', 9)])
    with pytest.raises(ValueError, match='DIFFERENT map'):
        G.resolve_views(s, wi[1:2].contiguous(), ori[1:2].contiguous(), view_ids=[('r', 0)])   # (a device batch tensor is looked at once per id)


def test_view_index_batch_and_file_checks(tmp_path):
    from nerfail_amd import GaussNet as G
    s, wi, ori = _toy()
    Ns = s.numel() // 4
    s2, wi2, _ = _toy(H=7, W=5)
    a, b = G.ViewIndex(wi[0], Ns), G.ViewIndex(wi2[0], s2.numel() // 4)
    with pytest.raises(ValueError, match='different image / table sizes'):
        G.view_table([a, b])
    # a persisted index of another table size next to the maps is refused by the loader
    d = tmp_path / 'maps'
    d.mkdir()
    torch.save(wi[0].cpu(), str(d / '0.pth'))
    b.save(str(d / '0.idx.pth'))
    with pytest.raises(ValueError):
        G.load_view_indices(str(d), [0], Ns, save_missing=False)


def test_step_and_logit_gradient_argument_checks():
    from nerfail_amd import GaussNet as G, attack as A
    s, wi, ori = _toy()
    with pytest.raises(ValueError, match='same shape'):
        A.igsm_step(s, s[:, :, :, :3].contiguous(), s)
    with pytest.raises(ValueError, match='3 floats per row'):
        A.igsm_step_rgb(s, torch.zeros(5, device=dev()), s)
    victim = torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d(2), torch.nn.Flatten(), torch.nn.Linear(12, 8)).to(dev()).requires_grad_(False)
    net = G.gauss_net(dev(), 0.02, victim, 'my_model')
    xr, cla, _, views, aux = net.attack_forward(s, wi, ori, None)
    cla.sum().backward()
    with pytest.raises(ValueError, match='rows of 4 floats'):
        G.hot_backward_rgb_step(aux, xr.grad, views, s[:1], s, 2.0, 32.0, False)      # a perturbation table of another size
    x, x_rgba, cla, _, _ = net(s.clone().requires_grad_(True), wi, ori)
    with pytest.raises(ValueError, match='one view at a time'):
        net.logit_gradients(s, wi, x, x_rgba, cla, [0, 1])                            # DeepFool runs at batch 1 (AN:82)
    x1, xr1, cla1, _, _ = net(s.clone().requires_grad_(True), wi[:1], ori[:1])
    with pytest.raises(ValueError, match='1..8 classes'):
        net.logit_gradients(s, wi[:1], x1, xr1, cla1, [])
    with pytest.raises(ValueError, match='B,2,H,W,8'):
        G.create_gauss_w(dev(), 0.02)(wi[:, :, :, :, :4])
    with pytest.raises(Exception, match='c must be positive'):
        G.create_gauss_w(dev(), 0.0)(wi)
    with pytest.raises(ValueError, match='B,2,H,W,8'):
        G.gauss_get_r(dev(), 0.02, victim, 'my_model')(s, wi[:, :, :, :, :4])
    with pytest.raises(ValueError, match=r'\[B,H,W,4\]'):
        G.gauss_get_img(dev(), 0.02, victim, 'my_model').compose(ori, ori[:1])


def test_nerf_mirror_refuses_what_the_hip_path_does_not_implement():
    from nerfail_amd import run_nerf as RN, run_nerf_helpers as RH
    from nerfail_amd.optim import Adam
    from nerfail_amd.run_nerf_helpers import NeRF
    _, net = hip_nerf(4, 64, 7)
    rays = T(synth.ray_batch(8, seed=1))
    focal, K = synth.lego_intrinsics(8, 8)
    c2w = synth.pose_spherical(10., -30., 4.)[:3, :4]
    kw = dict(network_fn=net, network_query_fn=None, N_samples=8, white_bkgd=True)
    with pytest.raises(NotImplementedError, match='ndc'):
        RN.render(8, 8, K, c2w=torch.from_numpy(c2w), ndc=True, near=2., far=6., use_viewdirs=True, **kw)
    with pytest.raises(NotImplementedError, match='use_viewdirs'):
        RN.render(8, 8, K, c2w=torch.from_numpy(c2w), ndc=False, near=2., far=6., use_viewdirs=False, **kw)
    with pytest.raises(NotImplementedError, match=r'\[R, 11\]'):
        RN.render_rays(rays[:, :8].contiguous(), net, None, 8)
    with pytest.raises(NotImplementedError, match='LLFF'):
        RH.ndc_rays(8, 8, 1.0, 1.0, rays[:, :3], rays[:, 3:6])
    with pytest.raises(NotImplementedError, match='one skip connection'):
        NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=5, skips=[2, 4], use_viewdirs=True).to(dev()).packed()
    with pytest.raises(NotImplementedError, match='use_viewdirs'):
        NeRF(D=4, W=64, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=False).to(dev()).packed()
    with pytest.raises(NotImplementedError, match='unsupported NeRF shape'):
        NeRF(D=4, W=96, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).to(dev()).packed()
    with pytest.raises(RuntimeError, match='no CPU path'):
        NeRF(D=4, W=64, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).packed()   # parameters on the CPU
    p = torch.nn.Parameter(torch.zeros(4, device=dev()))
    p.grad = torch.ones(4, device=dev())
    with pytest.raises(NotImplementedError, match='reference configuration'):
        Adam([p], lr=5e-4, weight_decay=0.1).step()                                   # RN:207 uses neither weight decay nor amsgrad
    p64 = torch.nn.Parameter(torch.zeros(4, dtype=torch.float64, device=dev()))
    p64.grad = torch.ones(4, dtype=torch.float64, device=dev())
    with pytest.raises(RuntimeError, match='contiguous float32'):
        Adam([p64], lr=5e-4).step()
    Adam([p], lr=5e-4).step()                                                         # the reference configuration steps
    assert float(p.detach().abs().max()) > 0


def test_knn_and_deepfool_argument_checks():
    from nerfail_amd import create_index_and_dist as CI
    from nerfail_amd.deepfool import deepfool
    q = T(synth.sphere_shell_points(16, seed=1).reshape(4, 4, 3))
    pts = T(synth.sphere_shell_points(64, seed=2))
    with pytest.raises(ValueError, match='auto, grid or brute'):
        CI.knn8(q, pts, method='kd-tree')
    with pytest.raises(Exception, match='8 points|8 <= M|unsupported point count'):
        CI.knn8(q, pts[:5], method='grid')
    with pytest.raises(NotImplementedError, match='universal_2d'):
        deepfool((None, None, None), 1.0, None, universal_2d=True)
