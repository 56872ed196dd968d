"""CPU checks of the Faster R-CNN oracle (oracle/rcnn_ops.py).  RoIAlign is PINNED by the reference's own known-answer test
(tests/layers/test_roi_pool.py:32-45: the 4x4 matrix; :64-75: invariance under a 2x bilinear upsampling with stride 1/2), re-typed
as data in tests/golden/reference_kat.npz.  The other operators have no reference vector (parity unpinned) and are pinned by the
properties their published definitions imply: RoIAlign reproduces constant and affine feature maps at the bin centres, its backward
is the exact adjoint, the FPN level rule maps the canonical 224-pixel box to level 4, and the samplers keep exactly the requested
number of smallest-key entries."""
import os

import numpy as np

from oracle import box_ops as ob
from oracle import rcnn_ops as orc

STRIDES = [4, 8, 16, 32]


def test_assign_roi_levels_known_values():
    def sq(s):
        return [10.0, 20.0, 10.0 + s, 20.0 + s]
    rois = np.array([sq(224), sq(223.9), sq(112), sq(111), sq(448), sq(2000), sq(8), sq(0)], np.float32)
    lv = orc.assign_roi_levels(rois, STRIDES)
    # floor(4 + log2(s / 224)) clamped to [2, 5], minus 2
    assert lv.tolist() == [2, 1, 1, 0, 3, 3, 0, 0]


def _kat():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.npz"))


def test_roi_align_reference_known_answer():
    """tests/layers/test_roi_pool.py:32-45: 5x5 arange feature, roi [0, 1, 1, 3, 3], strides [1], pool 4 -> the 4x4 matrix."""
    k = _kat()
    feat = k["roi_feat"].transpose(0, 2, 3, 1)                 # (1, 5, 5, 1) channel-last
    rois = k["roi_rois"]
    out = orc.roi_align([feat], rois[:, 1:], rois[:, 0].astype(int), [1], 4, 4, 2)
    assert np.allclose(out.reshape(4, 4), k["roi_align_4x4"])
    assert np.array_equal(out.reshape(4, 4).astype(np.float64), k["roi_align_4x4"])       # every value is exact in fp32


def test_roi_pool_max_reference_known_answer():
    """tests/layers/test_roi_pool.py:47-61."""
    k = _kat()
    out = orc.roi_pool_max(k["roi_feat"], k["roi_rois"], 1.0, 4, 4)
    assert np.array_equal(out[0, 0].astype(np.float64), k["roi_pool_4x4"])


def test_roi_align_reference_scale_invariance():
    """tests/layers/test_roi_pool.py:64-75: the same RoI on the 2x bilinearly upsampled map with strides [1/2] gives the same
    output (F.vision.interpolate defaults: bilinear, align_corners=False)."""
    import torch
    import torch.nn.functional as TF
    k = _kat()
    feat = k["roi_feat"]
    rois = k["roi_rois"]
    f2 = TF.interpolate(torch.from_numpy(feat), scale_factor=2, mode="bilinear", align_corners=False).numpy()
    out1 = orc.roi_align([feat.transpose(0, 2, 3, 1)], rois[:, 1:], [0], [1], 4, 4, 2)
    out2 = orc.roi_align([f2.transpose(0, 2, 3, 1)], rois[:, 1:], [0], [0.5], 4, 4, 2)
    assert np.allclose(out2, out1)


def test_roi_align_constant_and_affine_maps():
    rng = np.random.default_rng(0)
    H, W = 40, 60
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    feat = np.stack([np.full((H, W), 3.5, np.float32), 0.25 * yy + 0.5 * xx + 1.0], -1)[None]     # (1, H, W, 2)
    feats = [feat, feat[:, ::2, ::2], feat[:, ::4, ::4], feat[:, ::8, ::8]]
    # RoIs well inside the image, all on level 0 (side < 112): stride 4
    rois = np.array([[40.0, 32.0, 120.0, 100.0], [17.3, 21.9, 93.2, 77.7]], np.float32)
    assert orc.assign_roi_levels(rois, STRIDES).tolist() == [0, 0]
    out = orc.roi_align(feats, rois, [0, 0], STRIDES)
    assert np.allclose(out[..., 0], 3.5, atol=1e-6)
    for r, roi in enumerate(rois):
        x1, y1, x2, y2 = roi / 4.0 - 0.5                        # continuous feature coordinates (aligned=True)
        bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
        for ph in range(7):
            for pw in range(7):
                cy, cx = y1 + (ph + 0.5) * bh, x1 + (pw + 0.5) * bw       # mean of the 2x2 samples of an affine map
                assert abs(out[r, ph * 7 + pw, 1] - (0.25 * cy + 0.5 * cx + 1.0)) < 1e-4


def test_roi_align_backward_is_adjoint():
    rng = np.random.default_rng(1)
    N, C = 2, 3
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4)]
    feats = [rng.normal(0, 1, (N, h, w, C)).astype(np.float32) for h, w in sizes]
    rois = np.array([[5, 7, 60, 50], [-20, -20, 140, 130], [-10, -10, 30, 30], [50, 40, 58, 47], [-100, -80, 260, 200]], np.float32)
    bidx = np.array([0, 1, 1, 0, 1])
    out = orc.roi_align(feats, rois, bidx, STRIDES)
    g = rng.normal(0, 1, out.shape).astype(np.float32)
    grads = orc.roi_align_backward(g, [f.shape for f in feats], rois, bidx, STRIDES)
    lhs = float((out.astype(np.float64) * g).sum())
    rhs = sum(float((f.astype(np.float64) * gr).sum()) for f, gr in zip(feats, grads))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    assert len(set(orc.assign_roi_levels(rois, STRIDES).tolist())) >= 2


def test_topk_and_samplers():
    s = np.array([0.5, 2.0, -0.0, 0.0, 2.0, 1.0, -3.0], np.float32)
    idx, top = orc.topk_desc(s, 4)
    assert idx.tolist() == [1, 4, 5, 0]                       # ties: lower index first
    idx, _ = orc.topk_desc(s, 10, min_score=0.0)
    assert idx.tolist() == [1, 4, 5, 0]                       # strictly above the threshold
    idx, _ = orc.topk_desc(s, 7)
    assert idx.tolist()[4:6] == [3, 2]                        # +0.0 sorts above -0.0 (bit-pattern order)
    rng = np.random.default_rng(2)
    labels = rng.choice([-1, 0, 1], 500).astype(np.int32)
    keys = rng.random(500, dtype=np.float32)
    out = orc.sample_labels(labels, keys, 20, 1, -1)
    kept = np.nonzero(out == 1)[0]
    assert len(kept) == 20
    pos = np.nonzero(labels == 1)[0]
    assert set(kept) == set(pos[np.argsort(keys[pos], kind="stable")[:20]])
    assert np.array_equal(out[labels != 1], labels[labels != 1])
    assert np.array_equal(orc.sample_labels(labels, keys, 10 ** 6, 1, -1), labels)


def test_rpn_and_rcnn_ground_truth_invariants():
    rng = np.random.default_rng(3)
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    anchors = np.concatenate(ob.default_anchors(sizes, [4, 8, 16, 32, 64], [[x] for x in [32, 64, 128, 256, 512]], [[0.5, 1, 2]], 0.5), 0)
    gt = np.array([[[10, 12, 70, 80, 3], [40, 30, 120, 90, 7], [0, 0, 0, 0, 0]]], np.float32)
    A = len(anchors)
    kp, kn = rng.random((1, A), dtype=np.float32), rng.random((1, A), dtype=np.float32)
    labels, offsets = orc.rpn_ground_truth(anchors, gt, [2], kp, kn, num_sample_anchors=64, num_pos_anchor=32)
    assert (labels == 1).sum() <= 32 and (labels >= 0).sum() == 64
    assert (labels == 1).sum() >= 2                           # low-quality matches keep every gt
    rois = np.concatenate([gt[0, :2, :4] + rng.normal(0, 2, (2, 4)).astype(np.float32), rng.uniform(0, 90, (60, 4)).astype(np.float32)])
    rois[:, 2:] = np.maximum(rois[:, 2:], rois[:, :2] + 1)
    kf, kb = rng.random(100, dtype=np.float32), rng.random(100, dtype=np.float32)
    rr, rl, rt = orc.rcnn_ground_truth(rois, gt[0, :2], kf, kb, num_rois=16, fg_ratio=0.25)
    assert len(rl) == 16 and (rl > 0).sum() <= 4 and set(rl[rl > 0].tolist()) <= {3, 7}
    # a RoI identical to its gt has zero regression target
    r2, l2, t2 = orc.rcnn_ground_truth(np.zeros((0, 4), np.float32), gt[0, :2], kf, kb, num_rois=16)
    assert l2.tolist() == [3, 7] and np.allclose(t2, 0)


def test_detect_postprocess_small_case():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60]], np.float32)
    scores = np.zeros((3, 2), np.float32)
    scores[0, 1] = 0.9; scores[1, 1] = 0.8; scores[2, 0] = 0.7; scores[2, 1] = 0.04
    b, s, l = orc.detect_postprocess([scores.reshape(-1)], [boxes], 2, [100, 100, 200, 50, 0], 0.05, 0.5, 100)
    assert l.tolist() == [1, 0] and np.allclose(s, [0.9, 0.7])
    assert np.allclose(b, [[0, 0, 5, 20], [25, 100, 30, 120]])     # x * 50/100, y * 200/100, clipped to (200, 50)
