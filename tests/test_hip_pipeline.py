"""-m gpu: the NeRFail pipeline end to end THROUGH THE REFERENCE'S FILE CONTRACTS, at toy size:

    reference-format checkpoint (.tar, RN:210-225)  -> create_nerf
    -> nerf_to_coord.render_path -> renderonly_<split>_<epochs>/NNN.npy   (NC:172-173, [H,W,3] float32 pts_max)
    -> create_index_and_dist     -> index_and_dist/<split>/<i>.pth        (CI:148-163, float32 [2,H,W,8], idx as float)
    -> create_gauss_w            -> index_and_weight                      (DW:82-97)
    -> gauss_net + NeRFail-S step (AS:304-392)

Every stage's file is checked against the oracle computed from the PREVIOUS stage's file, so an integration error
(layout, dtype, index base, view order in the point set) cannot hide behind per-kernel parity."""
import os
import types

import numpy as np
import pytest
import torch

import synth
from conftest import rel_err
from hiputil import T, N, dev
from oracle import gauss as OG
from oracle import knn as OK

pytestmark = pytest.mark.gpu

H = W = 20


def _args(basedir):
    return types.SimpleNamespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64,
                                 netdepth=4, netwidth=64, netdepth_fine=4, netwidth_fine=64, netchunk=65536, lrate=5e-4,
                                 basedir=basedir, expname='blender_paper_toy', ft_path=None, no_reload=False, perturb=1.,
                                 white_bkgd=True, raw_noise_std=0., dataset_type='blender', no_ndc=False, lindisp=False)


def test_pipeline_through_files(tmp_path):
    from nerfail_amd import nerf_to_coord as NC
    from nerfail_amd.create_index_and_dist import create_index_and_dist
    from nerfail_amd.GaussNet import gauss_net
    from nerfail_amd.attack import nerfail_s_step
    base = str(tmp_path)
    logs = os.path.join(base, 'logs')
    exp = os.path.join(logs, 'blender_paper_toy')
    os.makedirs(exp)
    # ---- a checkpoint exactly as the reference writes it (RN:812-818), optimizer state from stock torch Adam on CPU
    sd_c, sd_f = synth.nerf_state_dict(D=4, W=64, seed=61), synth.nerf_state_dict(D=4, W=64, seed=62)
    cpu_params = [torch.nn.Parameter(torch.from_numpy(v.copy())) for sd in (sd_c, sd_f) for v in sd.values()]
    stock = torch.optim.Adam(cpu_params, lr=5e-4, betas=(0.9, 0.999))
    for p in cpu_params:
        p.grad = torch.zeros_like(p)
    stock.step()
    torch.save({'global_step': 7, 'network_fn_state_dict': {k: torch.from_numpy(v) for k, v in sd_c.items()},
                'network_fine_state_dict': {k: torch.from_numpy(v) for k, v in sd_f.items()},
                'optimizer_state_dict': stock.state_dict()}, os.path.join(exp, '000007.tar'))
    args = _args(logs)
    render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer = NC.create_nerf(args)
    assert start == 7 and float(optimizer.state[grad_vars[0]]['step']) == 1.0
    assert np.array_equal(N(render_kwargs_test['network_fn'].pts_linears[0].weight), sd_c['pts_linears.0.weight'])
    render_kwargs_test.update(near=2., far=6.)

    # ---- render_path -> NNN.npy per split
    focal, K = synth.lego_intrinsics(H, W)
    counts = {'test': 4, 'train': 2, 'val': 2}
    ang = {'test': 0., 'train': 90., 'val': 200.}
    for split, n in counts.items():
        d = os.path.join(exp, 'renderonly_%s_%06d' % (split, 7))
        os.makedirs(d)
        poses = torch.from_numpy(np.stack([synth.pose_spherical(ang[split] + 40. * i, -30., 4.) for i in range(n)]).astype(np.float32))
        with torch.no_grad():                                   # as the reference calls it (NC:640, RN:693)
            rgbs, disps = NC.render_path(poses, [H, W, focal], K, H * W, render_kwargs_test, savedir=d)
        assert rgbs.shape == (n, H, W, 3) and disps.shape == (n, H, W)
        for i in range(n):
            a = np.load(os.path.join(d, '%03d.npy' % i))
            assert a.shape == (H, W, 3) and a.dtype == np.float32 and np.isfinite(a).all()

    # ---- create_index_and_dist -> <i>.pth ; oracle on the .npy files (point set = views mask_list of `test`, in order)
    mask_list = [2, 0, 3]
    create_index_and_dist('toy', '%06d' % 7, mask_list, basedir=base, train_img_num=2, val_img_num=2, test_img_num=4)
    tdir = os.path.join(exp, 'renderonly_test_%06d' % 7)
    S = np.stack([np.load(os.path.join(tdir, '%03d.npy' % i)) for i in mask_list]).reshape(-1, 3)
    for split, n in counts.items():
        for i in range(n):
            got = torch.load(os.path.join(exp, 'index_and_dist', split, '%d.pth' % i))
            assert tuple(got.shape) == (2, H, W, 8) and got.dtype == torch.float32 and not got.is_cuda
            q = np.load(os.path.join(exp, 'renderonly_%s_%06d' % (split, 7), '%03d.npy' % i))
            want = OK.index_and_dist(q, S)
            assert np.array_equal(got[1].numpy(), want[1]), (split, i)            # indices (as float), exact
            assert np.array_equal(got[0].numpy(), want[0]), (split, i)            # distances, bit-exact
    # a view that is part of the point set finds itself at distance 0 with its own global index
    own = torch.load(os.path.join(exp, 'index_and_dist', 'test', '0.pth'))
    assert float(own[0][..., 0].max()) == 0.0
    assert np.array_equal(own[1][..., 0].numpy().reshape(-1), H * W * 1 + np.arange(H * W, dtype=np.float32))

    # ---- dist -> weight driver (DW:82-97) over all splits, then the batch of the 4 test views as the attack loads it
    from nerfail_amd.dist_to_weight import dist_to_weight
    v = dist_to_weight('toy', basedir=base, test_number=4, val_number=2, train_number=2, c=0.02)
    dai = torch.stack([torch.load(os.path.join(exp, 'index_and_dist', 'test', '%d.pth' % i)) for i in range(4)])
    files = torch.stack([torch.load(os.path.join(exp, 'index_and_weight', 'test', '%d.pth' % i)) for i in range(4)])
    assert tuple(files.shape) == (4, 2, H, W, 8) and files.dtype == torch.float32 and not files.is_cuda
    want, _ = OG.create_gauss_w(dai.numpy(), 0.02)
    assert rel_err(files.numpy()[:, 0], want[:, 0]) < 1e-6
    assert np.array_equal(files.numpy()[:, 1], dai.numpy()[:, 1])
    all_d = [torch.load(os.path.join(exp, 'index_and_dist', sp, '%d.pth' % i))[0] for sp, n in (('test', 4), ('val', 2), ('train', 2))
             for i in range(n)]
    assert abs(v - float(np.mean([float((d.double() ** 2).mean()) for d in all_d]))) < 1e-6 * max(v, 1e-12) + 1e-12
    wi = files.to(dev())

    # ---- two NeRFail-S steps over these maps with a toy victim
    torch.manual_seed(0)
    victim = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(2), torch.nn.Flatten(),
                                 torch.nn.Linear(16, 8)).to(dev())
    victim.requires_grad_(False)
    net = gauss_net(dev(), 0.02, victim, 'my_model', epsilon=None)
    ori = T(synth.disc_alpha_image(4, H, W, seed=4))
    s0 = torch.zeros((3, H, W, 4), device=dev())
    s0[..., 3] = 255.0
    s = s0.clone()
    for _ in range(2):
        s, loss = nerfail_s_step(net, s, s0, wi, ori, torch.tensor(3, device=dev()), a=2.0, epsilon=32.0)
    d = N(s - s0)
    assert np.array_equal(d[..., 3], np.zeros_like(d[..., 3])) and np.abs(d[..., :3]).max() <= 4.0 and np.abs(d).max() > 0
    assert np.isfinite(float(loss))

    # ---- the same two steps with the views' inverted indices stored next to their maps (N2): built and saved on the
    # first call, loaded from <i>.idx.pth on the second; the iterates are the same bits either way
    from nerfail_amd import GaussNet as G
    mdir = os.path.join(exp, 'index_and_weight', 'test')
    for rnd in range(2):
        G._VIEW_CACHE.clear(); G._BATCH_KEYS.clear()
        ids = G.load_view_indices(mdir, range(4), s0.numel() // 4)
        assert all(os.path.exists(os.path.join(mdir, '%d.idx.pth' % i)) for i in range(4))
        s2 = s0.clone()
        for _ in range(2):
            s2, _ = nerfail_s_step(net, s2, s0, wi.clone(), ori, torch.tensor(3, device=dev()), a=2.0, epsilon=32.0, view_ids=ids)
        assert torch.equal(s2, s)


def test_retrain_on_attacked_data_round_trip(tmp_path):
    """SURVEY 8f N3 - the paper's closing loop at toy size: a Blender-format scene on disk -> load_blender_data -> a few
    training steps (RN:746-801) -> adversarial versions of the TRAIN images written as PNGs (what attack_NeRFail_S.py's last
    epoch does, AS:394-403) -> load_blender_data(train_dir=...) (LB:62-63) -> training continues on the attacked images."""
    from PIL import Image
    from test_load_blender import write_toy_scene
    from nerfail_amd import run_nerf as RN
    from nerfail_amd.load_blender import load_blender_data, training_images, train_step
    root = str(tmp_path / 'scene')
    raw = write_toy_scene(root, H=16, W=16, n=(3, 2, 2), seed=3)
    images, poses, render_poses, hwf, i_split = load_blender_data(root)
    i_train = i_split[0]
    Hh, Ww, focal = hwf
    K = np.array([[focal, 0, 0.5 * Ww], [0, focal, 0.5 * Hh], [0, 0, 1]])
    logs = str(tmp_path / 'logs')
    os.makedirs(os.path.join(logs, 'blender_paper_toy'))
    args = _args(logs)
    torch.manual_seed(0)
    render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer = RN.create_nerf(args)
    assert start == 0
    with torch.no_grad():       # a freshly initialised NeRF has sigma <= 0 nearly everywhere (no density, no gradient):
        for net in (render_kwargs_train['network_fn'], render_kwargs_train['network_fine']):
            net.alpha_linear.bias += 0.5                       # the same nudge the fixtures use
    imgs = training_images(images, white_bkgd=True)
    rng = np.random.RandomState(0)
    w0 = N(grad_vars[0]).copy()
    losses = [train_step(imgs, poses, i_train, hwf, K, render_kwargs_train, optimizer, step, N_rand=128, rng=rng)[0]
              for step in range(1, 4)]
    assert all(np.isfinite(losses)) and np.abs(N(grad_vars[0]) - w0).max() > 0
    # "attack": a bounded perturbation of the train images, saved under the SAME file names in another directory
    adv_dir = str(tmp_path / 'adv_train')
    os.makedirs(adv_dir)
    adv = raw['train'].astype(np.int32)
    adv[..., :3] = np.clip(adv[..., :3] + np.random.RandomState(1).randint(-32, 33, adv[..., :3].shape), 0, 255)
    for i in range(3):
        Image.fromarray(adv[i].astype(np.uint8), 'RGBA').save(os.path.join(adv_dir, 'r_%d.png' % i))
    images2, poses2, _, hwf2, i_split2 = load_blender_data(root, train_dir=adv_dir)
    imgs2 = training_images(images2, white_bkgd=True, train_dir=adv_dir)
    assert imgs2.shape == imgs.shape and np.array_equal(imgs2[3:], imgs[3:]) and not np.array_equal(imgs2[:3], imgs[:3])
    rng = np.random.RandomState(0)
    l2, psnr, lr = train_step(imgs2, poses2, i_split2[0], hwf2, K, render_kwargs_train, optimizer, 4, N_rand=128, rng=rng)
    assert np.isfinite(l2) and np.isfinite(psnr) and 0 < lr <= 5e-4
    # the renderer still renders after the weight updates (packed images follow the parameter versions)
    with torch.no_grad():
        rgb, disp, acc, extras = RN.render(16, 16, K, chunk=256, c2w=torch.from_numpy(poses2[0, :3, :4]), near=2., far=6.,
                                           **{k: v for k, v in render_kwargs_test.items()})
    assert tuple(rgb.shape) == (16, 16, 3) and torch.isfinite(rgb).all()
