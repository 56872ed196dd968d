"""The reference's operator surface (basedet.layers / basedet.structures names and signatures) on the HIP kernels, through the `basedet`
import alias: losses against oracle/box_ops.py (float64 restatements of layers/losses/*.py) and torch-autograd of the same formulas
for the gradients, Matcher(matrix) / assign_rois / sample_labels bit-exact against the oracle, roi_pool on the reference's own known
answers (tests/layers/test_roi_pool.py:32-75), Boxes on tests/structures/test_boxes.py:72-94, the conv + FrozenBN fold on the exactness
rule of tests/layers/test_module_utils.py:41-49, and a training run started from a user-style config file through the
`basedet_train` entry."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import box_ops as ob
from oracle import rcnn_ops as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def test_elementwise_losses_match_the_oracle_and_autograd():
    from basedet.layers import binary_cross_entropy, sigmoid_focal_loss, smooth_l1_loss
    rng = np.random.default_rng(0)
    x = rng.normal(0, 3, (257, 80)).astype(np.float32)
    t = (rng.random((257, 80)) < 0.05).astype(np.float32)
    for alpha, gamma in ((0.25, 2.0), (-1, 0), (0.5, 1.5)):
        xt = _dev(x).requires_grad_(True)
        loss = sigmoid_focal_loss(xt, _dev(t), alpha=alpha, gamma=gamma)
        ref = ob.sigmoid_focal_loss(x.astype(np.float64), t.astype(np.float64), alpha, gamma)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), ref, rtol=2e-5, atol=1e-6)
        w = rng.normal(0, 1, x.shape).astype(np.float32)
        (loss * _dev(w)).sum().backward()
        # gradient: torch autograd of the reference's formula in float64 (sigmoid_focal_loss.py:29-35)
        x64 = torch.from_numpy(x).double().requires_grad_(True)
        t64 = torch.from_numpy(t).double()
        p = torch.sigmoid(x64)
        ce = -(t64 * torch.nn.functional.logsigmoid(x64) + (1 - t64) * torch.nn.functional.logsigmoid(-x64))
        l64 = ce
        if gamma != 0:
            l64 = l64 * (t64 * (1 - p) + (1 - t64) * p) ** gamma
        if alpha >= 0:
            l64 = l64 * (t64 * alpha + (1 - t64) * (1 - alpha))
        (l64 * torch.from_numpy(w).double()).sum().backward()
        np.testing.assert_allclose(xt.grad.cpu().numpy(), x64.grad.numpy(), rtol=2e-4, atol=2e-6)
    # binary cross entropy, logits and probabilities
    xt = _dev(x).requires_grad_(True)
    l = binary_cross_entropy(xt, _dev(t))
    np.testing.assert_allclose(l.detach().cpu().numpy(), ob.binary_cross_entropy(x.astype(np.float64), t.astype(np.float64)), rtol=2e-5, atol=1e-6)
    l.sum().backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), 1 / (1 + np.exp(-x.astype(np.float64))) - t, rtol=1e-4, atol=1e-6)
    pr = rng.uniform(0.05, 0.95, x.shape).astype(np.float32)
    l = binary_cross_entropy(_dev(pr), _dev(t), with_logits=False)
    np.testing.assert_allclose(l.cpu().numpy(), -(t * np.log(pr.astype(np.float64)) + (1 - t) * np.log(1 - pr.astype(np.float64))), rtol=2e-5)
    # smooth L1 (beta = 0 is plain L1, retinanet_cfg.py:32)
    a, b = rng.normal(0, 1, (300, 4)).astype(np.float32), rng.normal(0, 1, (300, 4)).astype(np.float32)
    for beta in (0.0, 0.11, 1.0):
        at = _dev(a).requires_grad_(True)
        l = smooth_l1_loss(at, _dev(b), beta=beta)
        np.testing.assert_allclose(l.detach().cpu().numpy(), ob.smooth_l1_loss(a.astype(np.float64), b.astype(np.float64), beta), rtol=1e-5, atol=1e-7)
        l.sum().backward()
        d = a.astype(np.float64) - b
        ref = np.sign(d) if beta < 1e-5 else np.where(np.abs(d) < beta, d / beta, np.sign(d))
        np.testing.assert_allclose(at.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def _ltrb_loss64(p, t, loss_type, eps=1e-8):
    """get_ltrb_boxes_iou + the loss map (iou_loss.py:9-56,95-100) in torch float64 (autograd gives the reference gradient)."""
    b1 = torch.cat([-p[..., :2], p[..., 2:]], -1)
    b2 = torch.cat([-t[..., :2], t[..., 2:]], -1)
    a1 = (b1[..., 2] - b1[..., 0]).clamp(min=0) * (b1[..., 3] - b1[..., 1]).clamp(min=0)
    a2 = (b2[..., 2] - b2[..., 0]).clamp(min=0) * (b2[..., 3] - b2[..., 1]).clamp(min=0)
    wi = (torch.minimum(b1[..., 2], b2[..., 2]) - torch.maximum(b1[..., 0], b2[..., 0])).clamp(min=0)
    hi = (torch.minimum(b1[..., 3], b2[..., 3]) - torch.maximum(b1[..., 1], b2[..., 1])).clamp(min=0)
    ai = wi * hi
    au = a1 + a2 - ai
    iou = ai / au.clamp(min=eps)
    if loss_type == "giou":
        gw = torch.maximum(b1[..., 2], b2[..., 2]) - torch.minimum(b1[..., 0], b2[..., 0])
        gh = torch.maximum(b1[..., 3], b2[..., 3]) - torch.minimum(b1[..., 1], b2[..., 1])
        ac = gw * gh
        iou = iou - (ac - au) / ac.clamp(min=eps)
    if loss_type == "iou":
        return -torch.log(iou.clamp(min=eps)), iou
    if loss_type == "square_iou":
        return 1 - iou ** 2, iou
    return 1 - iou, iou


def test_iou_loss_all_types_and_modes():
    from basedet.layers import iou_loss
    rng = np.random.default_rng(1)
    p = rng.uniform(0.5, 40, (500, 4)).astype(np.float32)
    t = rng.uniform(0.5, 40, (500, 4)).astype(np.float32)
    p[:5] = t[:5]                                      # identical boxes (iou = 1)
    for lt in ("iou", "linear_iou", "giou", "square_iou"):
        pt = _dev(p).requires_grad_(True)
        loss, ious = iou_loss(pt, _dev(t), box_mode="ltrb", loss_type=lt, return_iou=True)
        p64 = torch.from_numpy(p).double().requires_grad_(True)
        l64, i64 = _ltrb_loss64(p64, torch.from_numpy(t).double(), lt)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), l64.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ious.cpu().numpy(), i64.detach().numpy(), rtol=1e-4, atol=1e-5)
        if lt == "giou":                               # the numpy oracle restates this one as well
            np.testing.assert_allclose(loss.detach().cpu().numpy(), ob.iou_loss_ltrb(p.astype(np.float64), t.astype(np.float64), "giou"),
                                       rtol=1e-4, atol=1e-5)
        w = rng.normal(0, 1, 500).astype(np.float32)
        (loss * _dev(w)).sum().backward()
        (l64 * torch.from_numpy(w).double()).sum().backward()
        g, g64 = pt.grad.cpu().numpy(), p64.grad.numpy()
        ok = np.abs(p - t) > 1e-3                      # min/max kinks where pred == target are a convention
        np.testing.assert_allclose(g[ok], g64[ok], rtol=2e-3, atol=2e-5)
    # the other box modes: the PAIRWISE matrix of Boxes.iou / .giou mapped through the loss (iou_loss.py:83-100)
    b1 = np.array([[0, 0, 10, 10], [5, 5, 20, 25], [30, 30, 31, 31]], np.float32)
    b2 = np.array([[0, 0, 10, 10], [8, 8, 12, 30]], np.float32)
    for lt in ("iou", "linear_iou", "giou", "square_iou"):
        got = iou_loss(_dev(b1), _dev(b2), box_mode="xyxy", loss_type=lt).cpu().numpy()
        v = ob.box_giou(b1, b2) if lt == "giou" else ob.box_iou(b1, b2)
        ref = {"iou": -np.log(np.clip(v, 1e-8, None)), "square_iou": 1 - v ** 2}.get(lt, 1 - v)
        assert got.shape == (3, 2)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    xywh = np.concatenate([b1[:, :2], b1[:, 2:] - b1[:, :2]], 1)
    cxcywh = np.concatenate([(b1[:, :2] + b1[:, 2:]) / 2, b1[:, 2:] - b1[:, :2]], 1)
    ref = iou_loss(_dev(b1), _dev(b1), box_mode="xyxy", loss_type="giou").cpu().numpy()
    np.testing.assert_allclose(iou_loss(_dev(xywh), _dev(xywh), box_mode="xywh", loss_type="giou").cpu().numpy(), ref, atol=1e-6)
    np.testing.assert_allclose(iou_loss(_dev(cxcywh), _dev(cxcywh), box_mode="xcycwh", loss_type="giou").cpu().numpy(), ref, atol=1e-6)
    with pytest.raises(AssertionError):
        iou_loss(_dev(b1), _dev(b1), loss_type="ciou")


def test_matcher_on_a_matrix_is_bit_exact():
    from basedet.layers import Matcher
    from basedet.structures import Boxes
    rng = np.random.default_rng(2)
    anchors = ob.default_anchors([(20, 28), (10, 14)], [8, 16], [[32, 40, 50], [64, 80, 100]], [[0.5, 1, 2]], 0.5)
    anchors = np.concatenate(anchors, 0)
    gt = np.array([[10, 20, 90, 100], [100, 30, 200, 160], [5, 5, 30, 40], [300, 300, 310, 310]], np.float32)   # the last one overlaps nothing
    iou = ob.box_iou(gt, anchors)
    m = Boxes(_dev(gt)).iou(_dev(anchors))
    assert np.array_equal(m.cpu().numpy(), iou)
    for thr, labels, lq in (([0.4, 0.5], [0, -1, 1], True), ([0.3, 0.7], [0, -1, 1], False), ([0.5], [0, 1], False),
                            ([0.2, 0.4, 0.6], [0, -1, 2, 1], True)):
        idx, lab = Matcher(list(thr), labels, allow_low_quality_matches=lq)(m)
        ridx, rlab = ob.matcher(iou, list(thr), labels, lq)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(lab.cpu().numpy(), rlab), (thr, labels, lq)


def test_roi_pool_reference_known_answers(golden_dir):
    """tests/layers/test_roi_pool.py:32-61 (both 4x4 matrices) and :64-75 (2x upsampling at stride 1/2) through layers.roi_pool."""
    import torch.nn.functional as TF
    from basedet.layers import roi_pool
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    feat, rois = _dev(k["roi_feat"]), _dev(k["roi_rois"])
    out = roi_pool([feat], rois, strides=[1], pool_shape=4, pooler_type="roi_align")
    assert out.shape == (1, 1, 4, 4)
    assert np.allclose(out.cpu().numpy()[0, 0], k["roi_align_4x4"])
    outp = roi_pool([feat], rois, strides=[1], pool_shape=4, pooler_type="roi_pool")
    assert np.allclose(outp.cpu().numpy()[0, 0], k["roi_pool_4x4"])
    f2 = TF.interpolate(feat, scale_factor=2, mode="bilinear", align_corners=False)
    out2 = roi_pool([f2], rois, strides=[1 / 2], pool_shape=4, pooler_type="roi_align")
    assert np.allclose(out2.cpu().numpy(), out.cpu().numpy())


def test_roi_pool_multi_level_matches_the_oracle():
    from basedet.layers import assign_rois, roi_pool, sample_labels
    rng = np.random.default_rng(3)
    N, C = 2, 16
    sizes, strides = [(48, 80), (24, 40), (12, 20), (6, 10)], [4, 8, 16, 32]
    feats = [rng.normal(0, 1, (N, C, h, w)).astype(np.float32) for h, w in sizes]
    feats = [torch.from_numpy(f).to(torch.bfloat16).float().numpy() for f in feats]           # bf16-exact inputs
    R = 37
    b = np.stack([rng.uniform(0, 250, R), rng.uniform(0, 150, R)], 1)
    wh = rng.uniform(4, 300, (R, 2))
    boxes = np.concatenate([b, b + wh], 1).astype(np.float32)
    bidx = rng.integers(0, N, R)
    rois = np.concatenate([bidx[:, None].astype(np.float32), boxes], 1)
    out = roi_pool([_dev(f) for f in feats], _dev(rois), strides, (7, 7), "roi_align").cpu().numpy()      # (R, C, 7, 7)
    ref = orc.roi_align([f.transpose(0, 2, 3, 1) for f in feats], boxes, bidx, strides, 7, 7, 2)            # (R, 49, C)
    np.testing.assert_allclose(out, ref.reshape(R, 7, 7, C).transpose(0, 3, 1, 2), rtol=2 ** -7, atol=1e-3)
    r2, lv = assign_rois(_dev(rois), strides)
    assert r2.shape == (R + 4, 5) and np.array_equal(lv.cpu().numpy()[:R], orc.assign_roi_levels(boxes, strides))
    assert lv.cpu().numpy()[R:].tolist() == [0, 1, 2, 3] and float(r2[R:].abs().sum()) == 0
    # sample_labels with supplied keys == the oracle; without keys: the count rule
    labels = rng.integers(-1, 2, 5000).astype(np.int32)
    keys = rng.random(5000, dtype=np.float32)
    got = sample_labels(_dev(labels, torch.int32), 128, 1, keys=_dev(keys)).cpu().numpy()
    assert np.array_equal(got, orc.sample_labels(labels, keys, 128, 1))
    got = sample_labels(_dev(labels, torch.int32), 200, 0).cpu().numpy()
    assert (got == 0).sum() == 200 and np.array_equal(got == 1, labels == 1) and np.all(got[labels == -1] == -1)
    few = sample_labels(_dev(labels, torch.int32), 10 ** 6, 1).cpu().numpy()
    assert np.array_equal(few, labels)


def test_boxes_reference_known_answers(golden_dir):
    """tests/structures/test_boxes.py:72-74 (scale), :88-94 (__getitem__ types), plus filter_by_size / cat / clip."""
    from basedet.structures import Boxes
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    boxes1 = Boxes(_dev(k["boxes1"]))
    new_boxes = boxes1.scale(2, inplace=False)
    assert np.allclose(new_boxes.numpy(), boxes1.numpy() * 2) and np.allclose(boxes1.numpy(), k["boxes1"])
    sub = boxes1[:1]
    assert np.allclose(sub.numpy(), np.array([[0.0, 0.0, 1.0, 1.0]])) and isinstance(sub, Boxes)
    value = boxes1[0, 0]
    assert not isinstance(value, Boxes) and int(value.cpu().numpy()) == 0
    assert not isinstance(boxes1[:, 0], Boxes)
    b = Boxes(_dev(np.array([[0, 0, 10, 4], [0, 0, 3, 9], [-5, -5, 50, 50]], np.float32)))
    assert b.filter_by_size((5, 2)).cpu().tolist() == [False, True, True]           # (height, width) thresholds
    assert b.filter_by_size(3).cpu().tolist() == [True, False, True]
    c = b.cat(boxes1, inplace=False)
    assert isinstance(c, Boxes) and c.shape == (5, 4)
    clipped = b.clip((20, 30), inplace=False)
    assert clipped.numpy()[2].tolist() == [0, 0, 30, 20] and b.numpy()[2].tolist() == [-5, -5, 50, 50]
    b.scale((2, 3))
    assert b.numpy()[0].tolist() == [0, 0, 30, 8]


def test_conv_frozen_bn_fold_is_exact_without_eps():
    """tests/layers/test_module_utils.py:41-49: with eps = 0 and a fresh BN (gamma 1, beta 0, mean 0, var 1) the fused convolution
    returns exactly the outputs of conv-then-BN.  Here the fold is bd_weight_pack's row_scale (gamma / sqrt(var + eps)) plus the
    epilogue shift: identity statistics must leave every packed weight and every output bit unchanged, and a power-of-two scale
    (exact in bf16) must commute with the convolution bit for bit."""
    from basedet_amd import ops
    from tests.util import nchw_to_pm, pack_weights
    g = torch.Generator().manual_seed(0)
    N, Cin, Cout, H, W = 1, 8, 8, 10, 10
    x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).float()
    x[:, 3:] = 0                                                         # the reference's 3 input channels, padded to 8
    w = torch.randn(Cout, Cin, 1, 1, generator=g)
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    xp = nchw_to_pm(x)

    def run(scale, shift):
        wf, _ = pack_weights(ops, w, scale)
        y = torch.empty((N * H * W, Cout), dtype=torch.bfloat16, device="cuda")
        ops.conv2d_fwd(d, xp, wf, None if shift is None else shift.cuda(), y)
        return wf, y

    wf0, y0 = run(None, None)
    gamma, beta, mean, var, eps = torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout), 0.0
    scale = gamma / torch.sqrt(var + eps)
    wf1, y1 = run(scale, beta - mean * scale)
    assert torch.equal(wf0, wf1) and torch.equal(y0, y1)
    scale2 = torch.tensor([2.0, 0.5, 4.0, 1.0, 0.25, 8.0, 2.0, 1.0])
    _, y2 = run(scale2, torch.zeros(Cout))
    assert torch.equal(y2.float(), y0.float() * scale2.cuda())


def test_module_wrappers_agree_with_the_model_forward():
    from basedet.configs import FCOSConfig, RetinaNetConfig
    from basedet.layers import FPN, PointHead, RetinaNetHead
    from basedet.models import FCOS, RetinaNet
    from basedet_amd.configs import retinanet_r18_config
    from basedet_amd.models import params as P
    from basedet_amd.utils import DummyLoader
    cfg = retinanet_r18_config()
    cfg.MODEL.BATCHSIZE = 2
    model = RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0))
    b = next(DummyLoader(2, (128, 160), seed=0))
    img = (b["data"] * 255).astype(np.float32)
    feats = FPN(cfg, model=model)(img)
    assert list(feats) == ["p3", "p4", "p5", "p6", "p7"] and feats["p3"].shape == (2, 256, 16, 20) and feats["p7"].shape == (2, 256, 1, 2)
    logits, offsets = RetinaNetHead(cfg, model=model)([feats[k] for k in feats])
    assert logits[0].shape == (2, 720, 16, 20) and offsets[4].shape == (2, 36, 1, 2)
    # the same numbers as the model's own forward (features pass through bf16 both ways)
    model.train()
    model({k: b[k] if k != "data" else img for k in b})
    pl = model._cur
    from basedet_amd.layers.modules import _level
    for i in range(5):
        assert torch.equal(_level(pl.logits, pl.pyr, i), logits[i])
    fc = FCOSConfig()
    fc.merge(dict(MODEL=dict(BACKBONE=dict(NAME="resnet18", OUT_FEATURE_CHANNELS=[128, 256, 512]), FPN=dict(TOP_BLOCK_IN_CHANNELS=512))))
    fm = FCOS(fc, params=P.init_fcos_params(fc, seed=0))
    lg, off, ctr = PointHead(fc, model=fm)([feats[k] for k in feats])
    assert lg[0].shape == (2, 80, 16, 20) and off[1].shape == (2, 4, 8, 10) and ctr[2].shape == (2, 1, 4, 5)
    assert float(torch.stack([o.min() for o in off]).min()) >= 0.0          # relu(x * scale) * stride (point_head.py:143)


def test_retinanet_head_module_backward_matches_autograd():
    """layers/head/retina_head.py:103-112 as a composable module WITH gradients: RetinaNetHead(features) -> (logits, offsets), then
    head.backward(d_logits, d_offsets) -> d_features and head.grads() -> parameter gradients, against torch autograd of the same
    convolution stack (fp32 on the CPU, tower activations rounded to bf16 like the kernels' stored ones) fed the bf16-rounded features:
    outputs <= 2e-2, feature and parameter gradients <= 3e-2 rel-L2."""
    import torch.nn.functional as TF
    from basedet.layers import RetinaNetHead
    from basedet_amd.configs import retinanet_r18_config
    from basedet_amd.models import RetinaNet, params as P
    cfg = retinanet_r18_config()
    cfg.MODEL.BATCHSIZE = 2
    params = P.init_retinanet_params(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    for k in list(params):
        if k.startswith("head.") and k.endswith(".weight"):
            params[k] = (torch.randn(params[k].shape, generator=g) * 0.05).numpy()
    model = RetinaNet(cfg, params=params)
    N, sizes = 2, [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    feats = [torch.randn(N, 256, h, w, generator=g).bfloat16().float() for h, w in sizes]
    head = RetinaNetHead(cfg, model=model)
    logits, offsets = head([f.cuda() for f in feats])
    d_logits = [torch.randn(N, 720, h, w, generator=g).bfloat16().float() * 0.01 for h, w in sizes]
    d_offsets = [torch.randn(N, 36, h, w, generator=g).bfloat16().float() * 0.01 for h, w in sizes]
    d_feats = head.backward([t.cuda() for t in d_logits], [t.cuda() for t in d_offsets])
    grads = head.grads()
    # reference: the same stack in fp32 with autograd
    W = {k: torch.tensor(v, dtype=torch.float32).bfloat16().float().requires_grad_(True) if k.endswith(".weight")
         else torch.tensor(v, dtype=torch.float32).requires_grad_(True) for k, v in params.items() if k.startswith("head.")}
    fr = [f.clone().requires_grad_(True) for f in feats]
    loss = 0.0
    ref_logits, ref_offsets = [], []

    def bf(t):          # the kernels store tower activations as bf16: round the reference's alike (straight-through), so the ReLU gates agree
        return t + (t.bfloat16().float() - t).detach()
    for f, dl, do in zip(fr, d_logits, d_offsets):
        c, b = f, f
        for i in (0, 2, 4, 6):
            c = bf(TF.relu(TF.conv2d(c, W[f"head.cls_subnet.{i}.weight"], W[f"head.cls_subnet.{i}.bias"], padding=1)))
            b = bf(TF.relu(TF.conv2d(b, W[f"head.bbox_subnet.{i}.weight"], W[f"head.bbox_subnet.{i}.bias"], padding=1)))
        lg = TF.conv2d(c, W["head.cls_score.weight"], W["head.cls_score.bias"], padding=1)
        of = TF.conv2d(b, W["head.bbox_pred.weight"], W["head.bbox_pred.bias"], padding=1)
        ref_logits.append(lg); ref_offsets.append(of)
        loss = loss + (lg * dl).sum() + (of * do).sum()
    loss.backward()

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for i in range(5):
        assert rel(logits[i].cpu(), ref_logits[i].detach()) < 2e-2 and rel(offsets[i].cpu(), ref_offsets[i].detach()) < 2e-2, i
        assert rel(d_feats[i].cpu(), fr[i].grad) < 3e-2, (i, rel(d_feats[i].cpu(), fr[i].grad))
    assert set(grads) == set(W), sorted(set(grads) ^ set(W))
    for k, w in W.items():
        assert rel(grads[k].cpu().reshape(w.shape), w.grad) < 3e-2, (k, rel(grads[k].cpu().reshape(w.shape), w.grad))


_USER_CFG = '''
from basedet.configs import RetinaNetConfig


class Cfg(RetinaNetConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.MODEL.BACKBONE.NAME = "resnet18"
        self.MODEL.BACKBONE.OUT_FEATURE_CHANNELS = [128, 256, 512]
        self.MODEL.FPN.TOP_BLOCK_IN_CHANNELS = 512
        self.MODEL.BATCHSIZE = 2
        self.DATA.DUMMY_SIZE = (128, 160)
        self.GLOBAL.LOG_INTERVAL = 1
'''


def test_basedet_train_entry_runs_a_user_config(tmp_path):
    """tools/det_train.py:117-131: `basedet_train -f config.py` imports the file, instantiates Cfg, builds the trainer and trains."""
    import subprocess
    path = tmp_path / "config.py"
    path.write_text(_USER_CFG)
    base = [sys.executable, "-m", "basedet.tools.det_train", "-f", str(path), "--iters", "3"]
    # without an explicit opt-in the entry refuses to train a COCO-reader config on synthetic noise (this build has no dataset readers)
    r = subprocess.run(base + ["SOLVER.WARM_ITERS", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode != 0 and "DummyLoader" in r.stderr, (r.returncode, r.stderr[-2000:])
    r = subprocess.run(base + ["--synthetic", "SOLVER.WARM_ITERS", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-1000:])
    lines = [l for l in r.stdout.splitlines() if "total_loss" in l]
    assert len(lines) == 3 and "iter 3/" in lines[-1], r.stdout[-2000:]
    loss = float(lines[-1].split("total_loss")[1].split()[0])
    assert np.isfinite(loss)
