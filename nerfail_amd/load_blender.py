"""Mirror of Create_spatial_point_set/nerf_pytorch/load_blender.py (reference = LB): the Blender-synthetic loader with the
NeRFail-specific `train_dir` override (LB:37, :62-63, :69-73, :107-108) that swaps the attacked training images in for the
originals when a NeRF is retrained on adversarial data (README "retrain", RN:573-596).

Host-side file I/O: PNG decoding through imageio when present (as the reference), else PIL;
`half_res` is the reference's cv2.INTER_AREA at factor 2 = the mean of each 2x2 block. Same return structure:
    imgs, poses, render_poses, [H, W, focal], i_split               (train_dir is None)
    [train_imgs, imgs], poses, render_poses, [H, W, focal], i_split (train_dir given: `imgs` then holds val + test only,
                                                                     while i_split still counts the train views first)
`training_images()` is the RN:573-596 glue that turns either form into the [N,H,W,3] array train() samples from, and
`train_step()` one iteration of the RN:746-801 loop over the HIP render path."""
import json
import os

import numpy as np
import torch


def _imread(path):
    try:
        import imageio
        return np.asarray(imageio.imread(path))
    except ImportError:
        from PIL import Image
        return np.asarray(Image.open(path))


def pose_spherical(theta, phi, radius):
    """LB:29-34 (float32 torch matrices, same product order)."""
    t = torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]).float()
    ph = phi / 180. * np.pi
    rp = torch.Tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]]).float()
    th = theta / 180. * np.pi
    rt = torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]]).float()
    c2w = rt @ (rp @ t)
    return torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])) @ c2w


def _half(imgs):
    """cv2.resize(img, (W//2, H//2), INTER_AREA) for every image: the 2x2 block mean (odd trailing row / column dropped)."""
    n, H, W, C = imgs.shape
    h, w = H // 2, W // 2
    return imgs[:, :2 * h, :2 * w].reshape(n, h, 2, w, 2, C).mean((2, 4))


def load_blender_data(basedir, half_res=False, testskip=1, train_dir=None):
    """LB:37-110."""
    splits = ['train', 'val', 'test']
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, 'transforms_{}.json'.format(s)), 'r') as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, train_imgs = [], [], []
    counts = [0]
    for s in splits:
        meta = metas[s]
        imgs, poses = [], []
        skip = 1 if (s == 'train' or testskip == 0) else testskip
        for frame in meta['frames'][::skip]:
            fname = os.path.join(basedir, frame['file_path'] + '.png')
            if s == 'train' and train_dir is not None:          # LB:62-63: the attacked image of the same name
                fname = os.path.join(train_dir, os.path.basename(fname))
            imgs.append(_imread(fname))
            poses.append(np.array(frame['transform_matrix']))
        imgs = (np.array(imgs) / 255.).astype(np.float32)        # keeps all 4 channels (RGBA)
        poses = np.array(poses).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        if s == 'train' and train_dir is not None:
            train_imgs.append(imgs)
        else:
            all_imgs.append(imgs)
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    train_np_imgs = np.concatenate(train_imgs, 0) if train_dir is not None else None
    H, W = imgs[0].shape[:2]
    camera_angle_x = float(meta['camera_angle_x'])
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    render_poses = torch.stack([pose_spherical(angle, -30.0, 4.0) for angle in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        if train_dir is not None:
            train_np_imgs = _half(train_np_imgs).astype(np.float64)     # (the reference fills np.zeros: float64)
        imgs = _half(imgs).astype(np.float64)
    if train_dir is not None:
        return [train_np_imgs, imgs], poses, render_poses, [H, W, focal], i_split
    return imgs, poses, render_poses, [H, W, focal], i_split


def training_images(images, white_bkgd, train_dir=None):
    """RN:573-596: RGBA -> RGB (composited on white or alpha dropped), the attacked train images in front when `train_dir`."""
    train_images = None
    if train_dir is not None:
        train_images, images = images

    def rgb(a):
        if a.shape[3] <= 3:
            return a
        return a[..., :3] * a[..., -1:] + (1. - a[..., -1:]) if white_bkgd else a[..., :3]
    images = rgb(images)
    if train_dir is not None:
        images = np.concatenate([rgb(train_images), images], axis=0)
    return images


def train_step(images, poses, i_train, hwf, K, render_kwargs_train, optimizer, global_step, N_rand=1024, lrate=5e-4,
               lrate_decay=250, chunk=1024 * 32, precrop_iters=0, precrop_frac=.5, near=2., far=6., rng=np.random):
    """One iteration of the no_batching loop RN:746-801: random train image, get_rays on the full image (RN:752), N_rand
    random pixels (RN:768), render with gradients, loss = mse(rgb) + mse(rgb0), backward, Adam step, lr decay.
    Returns (loss, psnr, new_lrate)."""
    from . import run_nerf as RN
    from .run_nerf_helpers import get_rays, img2mse, mse2psnr
    H, W, focal = hwf
    H, W = int(H), int(W)
    dev = next(render_kwargs_train['network_fn'].parameters()).device
    img_i = rng.choice(i_train)
    target = torch.Tensor(images[img_i]).to(dev)
    pose = poses[img_i, :3, :4]
    rays_o, rays_d = get_rays(H, W, K, torch.Tensor(pose))
    if global_step < precrop_iters:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        coords = torch.stack(torch.meshgrid(torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH),
                                            torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW), indexing='ij'), -1)
    else:
        coords = torch.stack(torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing='ij'), -1)
    coords = torch.reshape(coords, [-1, 2])
    select_inds = rng.choice(coords.shape[0], size=[min(N_rand, coords.shape[0])], replace=False)
    select_coords = coords[select_inds].long().to(rays_o.device)
    rays_o = rays_o[select_coords[:, 0], select_coords[:, 1]]
    rays_d = rays_d[select_coords[:, 0], select_coords[:, 1]]
    batch_rays = torch.stack([rays_o, rays_d], 0)
    target_s = target[select_coords[:, 0], select_coords[:, 1]]
    rgb, disp, acc, extras = RN.render(H, W, K, chunk=chunk, rays=batch_rays, near=near, far=far, retraw=True, **render_kwargs_train)
    optimizer.zero_grad()
    img_loss = img2mse(rgb, target_s)
    loss = img_loss
    psnr = mse2psnr(img_loss.detach())
    if 'rgb0' in extras:
        loss = loss + img2mse(extras['rgb0'], target_s)
    loss.backward()
    optimizer.step()
    new_lrate = lrate * (0.1 ** (global_step / (lrate_decay * 1000)))            # RN:796-800
    for param_group in optimizer.param_groups:
        param_group['lr'] = new_lrate
    return float(loss.detach()), float(psnr), new_lrate
