"""Deterministic inputs of the multi-rank product-path tests (numpy only: the rank scripts on the GPU and the
oracle check on the CPU rebuild identical arrays from the seeds)."""
import numpy as np

import synth

P, B, H, W = 3, 5, 24, 24            # 5 views over 2 ranks: ragged split 3 + 2
LABEL, A, EPS, ITERS = 4, 2.0, 32.0, 3
RH_, RW_ = 40, 40                    # sharded render: 1600 rays over 2 ranks


def attack_inputs():
    rs = np.random.RandomState(7)
    s0 = np.zeros((P, H, W, 4), np.float32)
    s0[..., 3] = np.where(rs.uniform(size=(P, H, W)) < 0.85, 255.0, 0.0)
    ori = synth.disc_alpha_image(B, H, W, seed=8)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(B, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, (B, H, W, 8)).astype(np.float32)
    # images are 0..255: the logits must stay O(1). (Until round 6 the scale was 0.05: logit 125 on the labelled class, a
    # saturated softmax, CE = 0 and a perturbation gradient of 2e-39 - every gradient comparison built on it was vacuous.)
    cls_w = (rs.normal(size=(8, 3 * 4 * 4)) * 1e-3).astype(np.float32)
    return dict(s0=s0, ori=ori, dist_and_index=np.stack([dist, idx], 1), cls_w=cls_w)


def render_inputs():
    focal, K = synth.lego_intrinsics(RH_, RW_)
    c2w = synth.pose_spherical(-60.0, -30.0, 4.0)[:3, :4]
    return dict(K=K, c2w=c2w, seed_coarse=51, seed_fine=52)


# cfg5's loop shape (BASELINE.json configs[4]: attack_NeRFail_S over many views for many iterations at 1/2/4/8 GPUs),
# scaled down: 9 views in 3 batches of 3 (each batch split 2 + 1 over two ranks), 4 iterations, views named by dataset id
LOOP_VIEWS, LOOP_BATCH, LOOP_ITERS = 9, 3, 4


def loop_inputs():
    rs = np.random.RandomState(17)
    s0 = np.zeros((P, H, W, 4), np.float32)
    s0[..., 3] = np.where(rs.uniform(size=(P, H, W)) < 0.85, 255.0, 0.0)
    ori = synth.disc_alpha_image(LOOP_VIEWS, H, W, seed=18)
    dist = np.sort(np.abs(rs.normal(scale=0.02, size=(LOOP_VIEWS, H, W, 8))).astype(np.float32), -1)
    idx = rs.randint(0, P * H * W, (LOOP_VIEWS, H, W, 8)).astype(np.float32)
    return dict(s0=s0, ori=ori, dist_and_index=np.stack([dist, idx], 1))
