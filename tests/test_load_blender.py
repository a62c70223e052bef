"""CPU: the Blender loader mirror (load_blender.py:37-110) on a toy scene written in the dataset's own format, with and
without the NeRFail `train_dir` override: (i) against what the REFERENCE's load_blender_data returned for the same files
(fixture g18, tests/golden/make_golden.py), every array compared exactly; (ii) against expectations derived from the
files themselves (incl. half_res, which the fixture cannot pin: the reference needs cv2.resize, absent in the container)."""
import json
import os

import numpy as np
import pytest

import synth


def write_toy_scene(root, H=12, W=12, n=(3, 2, 2), seed=0):
    """transforms_{train,val,test}.json + RGBA PNGs in the Blender-synthetic layout; returns the raw uint8 images."""
    from PIL import Image
    rs = np.random.RandomState(seed)
    raw = {}
    for split, cnt in zip(('train', 'val', 'test'), n):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames, imgs = [], []
        for i in range(cnt):
            img = rs.randint(0, 256, size=(H, W, 4)).astype(np.uint8)
            img[..., 3] = np.where(rs.uniform(size=(H, W)) < 0.6, 255, 0)
            Image.fromarray(img, 'RGBA').save(os.path.join(root, split, 'r_%d.png' % i))
            pose = synth.pose_spherical(40. * i + {'train': 0., 'val': 13., 'test': 27.}[split], -30., 4.)
            frames.append({'file_path': './%s/r_%d' % (split, i), 'rotation': 0.1, 'transform_matrix': pose.tolist()})
            imgs.append(img)
        json.dump({'camera_angle_x': synth.LEGO_CAMERA_ANGLE_X, 'frames': frames},
                  open(os.path.join(root, 'transforms_%s.json' % split), 'w'))
        raw[split] = np.stack(imgs)
    return raw


def test_load_blender_data_plain_and_train_dir(tmp_path):
    from PIL import Image
    from nerfail_amd.load_blender import load_blender_data, training_images
    root = str(tmp_path / 'toy')
    raw = write_toy_scene(root)
    imgs, poses, render_poses, hwf, i_split = load_blender_data(root, half_res=False, testskip=1)
    assert imgs.shape == (7, 12, 12, 4) and imgs.dtype == np.float32 and poses.shape == (7, 4, 4)
    assert [list(s) for s in i_split] == [[0, 1, 2], [3, 4], [5, 6]]
    assert np.array_equal(imgs[:3], (raw['train'] / 255.).astype(np.float32)) and np.array_equal(imgs[5:], (raw['test'] / 255.).astype(np.float32))
    assert hwf[0] == 12 and hwf[1] == 12 and abs(hwf[2] - .5 * 12 / np.tan(.5 * synth.LEGO_CAMERA_ANGLE_X)) < 1e-9
    assert tuple(render_poses.shape) == (40, 4, 4)
    assert np.allclose(render_poses[0].numpy(), synth.pose_spherical(-180., -30., 4.), atol=1e-6)
    assert np.allclose(poses[0], synth.pose_spherical(0., -30., 4.), atol=1e-6)
    # testskip thins val / test only (LB:54-57)
    assert load_blender_data(root, testskip=2)[0].shape[0] == 3 + 1 + 1
    # half_res = 2x2 block mean (cv2.INTER_AREA at factor 2), focal halved
    h_imgs, _, _, h_hwf, _ = load_blender_data(root, half_res=True)
    assert h_imgs.shape == (7, 6, 6, 4) and h_hwf[:2] == [6, 6] and abs(h_hwf[2] - hwf[2] / 2) < 1e-12
    assert np.allclose(h_imgs[0, 1, 2], imgs[0, 2:4, 4:6].reshape(4, 4).mean(0), atol=1e-6)

    # ---- train_dir: attacked images of the same file names replace the train split (LB:62-63, :69-73, :107-108)
    adv = str(tmp_path / 'adv')
    os.makedirs(adv)
    adv_raw = 255 - raw['train']
    adv_raw[..., 3] = raw['train'][..., 3]
    for i in range(3):
        Image.fromarray(adv_raw[i], 'RGBA').save(os.path.join(adv, 'r_%d.png' % i))
    (t_imgs, rest), poses2, _, _, i_split2 = load_blender_data(root, train_dir=adv)
    assert t_imgs.shape == (3, 12, 12, 4) and rest.shape == (4, 12, 12, 4)            # `imgs` now holds val + test only
    assert np.array_equal(t_imgs, (adv_raw / 255.).astype(np.float32)) and np.array_equal(rest, imgs[3:])
    assert np.array_equal(poses2, poses) and [list(s) for s in i_split2] == [list(s) for s in i_split]
    # RN:573-596 glue: white background composite, attacked train images first -> indexable by i_split again
    tr = training_images([t_imgs, rest], white_bkgd=True, train_dir=adv)
    assert tr.shape == (7, 12, 12, 3)
    want0 = t_imgs[0, ..., :3] * t_imgs[0, ..., 3:] + (1. - t_imgs[0, ..., 3:])
    assert np.allclose(tr[0], want0) and np.allclose(tr[3], imgs[3, ..., :3] * imgs[3, ..., 3:] + (1. - imgs[3, ..., 3:]))
    assert training_images(imgs, white_bkgd=False).shape == (7, 12, 12, 3)
    with pytest.raises(FileNotFoundError):
        load_blender_data(root, train_dir=str(tmp_path / 'missing'))


def test_load_blender_matches_reference_fixture(tmp_path):
    """g18: the reference's load_blender_data was RUN on this very scene (files rebuilt here from the fixture's raw uint8
    images and JSON texts; PNG is lossless). Every returned array must be equal, dtype included."""
    from PIL import Image
    from nerfail_amd.load_blender import load_blender_data, pose_spherical
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g18_load_blender.npz'))
    root = str(tmp_path / 'toy')
    for split in ('train', 'val', 'test'):
        os.makedirs(os.path.join(root, split))
        open(os.path.join(root, 'transforms_%s.json' % split), 'wb').write(g['json_' + split].tobytes())
        for i, img in enumerate(g['raw_' + split]):
            Image.fromarray(img, 'RGBA').save(os.path.join(root, split, 'r_%d.png' % i))
    adv = str(tmp_path / 'adv')
    os.makedirs(adv)
    for i, img in enumerate(g['raw_adv']):
        Image.fromarray(img, 'RGBA').save(os.path.join(adv, 'r_%d.png' % i))

    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), (a.dtype, b.dtype, a.shape, b.shape)

    imgs, poses, render_poses, hwf, i_split = load_blender_data(root, half_res=False, testskip=1)
    same(imgs, g['plain_imgs']); same(poses, g['plain_poses']); same(render_poses.numpy(), g['plain_render_poses'])
    same(np.array(hwf, np.float64), g['plain_hwf']); same(np.concatenate(i_split), g['plain_i_split'])
    assert [len(s) for s in i_split] == list(g['plain_i_split_sizes'])
    assert isinstance(hwf[0], int) and isinstance(hwf[1], int)
    imgs2, poses2, _, _, i_split2 = load_blender_data(root, half_res=False, testskip=2)
    same(imgs2, g['skip2_imgs']); same(poses2, g['skip2_poses'])
    assert [len(s) for s in i_split2] == list(g['skip2_i_split_sizes'])
    (t_imgs, rest), poses3, rp3, hwf3, i_split3 = load_blender_data(root, train_dir=adv)
    same(t_imgs, g['adv_train_imgs']); same(rest, g['adv_rest_imgs']); same(poses3, g['adv_poses'])
    same(rp3.numpy(), g['adv_render_poses']); same(np.array(hwf3, np.float64), g['adv_hwf'])
    same(np.concatenate(i_split3), g['adv_i_split'])
    assert [len(s) for s in i_split3] == list(g['adv_i_split_sizes'])
    same(pose_spherical(37.0, -30.0, 4.0).numpy(), g['pose_spherical_37_m30_4'])
