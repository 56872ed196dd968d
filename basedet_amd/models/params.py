"""Parameter tables (names, shapes, initialisers) of the detectors, as plain numpy dicts.

Names follow the reference's state_dict (``backbone.bottom_up.layer2.0.conv1.weight`` ...), conv weights are
OIHW float32, biases / BN vectors are 1-D.  Initialisers restate:
  * ResNet: msra_normal fan_out for convs, BN gamma=1 beta=0 mean=0 var=1 (models/cls/resnet.py:174-189)
  * FPN lateral/output: msra_normal fan_in, zero bias (layers/backbone/fpn_backbone.py:78-83)
  * P6/P7: MegEngine ``M.Conv2d`` default init N(0, sqrt(1/fan_in)), zero bias (fpn_backbone.py:196-197)
  * RetinaNetHead: N(0, 0.01), zero bias, cls_score bias = -log((1-pi)/pi) (layers/head/retina_head.py:114-126)
"""
import math

import numpy as np

RESNET_SPECS = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
    "resnet152": ("bottleneck", [3, 8, 36, 3]),
}


def resnet_conv_table(name):
    """[(prefix, cin, cout, k, stride, pad, bn_prefix)] in forward order + per-block structure."""
    kind, layers = RESNET_SPECS[name]
    exp = 4 if kind == "bottleneck" else 1
    blocks = []
    cin = 64
    for li, nblk in enumerate(layers):
        ch = 64 * 2 ** li
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 0) else 1
            pre = f"backbone.bottom_up.layer{li + 1}.{b}"
            has_ds = (cin != ch * exp) or stride != 1
            blocks.append(dict(prefix=pre, kind=kind, cin=cin, ch=ch, cout=ch * exp, stride=stride, has_ds=has_ds, layer=li + 1))
            cin = ch * exp
    return blocks


def _msra_normal(rng, shape, mode):
    co, ci, kh, kw = shape
    fan = co * kh * kw if mode == "fan_out" else ci * kh * kw
    return (rng.standard_normal(shape) * math.sqrt(2.0 / fan)).astype(np.float32)


def _bn(p, prefix, c):
    p[prefix + ".weight"] = np.ones(c, np.float32)
    p[prefix + ".bias"] = np.zeros(c, np.float32)
    p[prefix + ".running_mean"] = np.zeros(c, np.float32)
    p[prefix + ".running_var"] = np.ones(c, np.float32)


def init_resnet(p, rng, name, residual_gamma=None):
    """residual_gamma: optional FrozenBN gamma for the last BN of every residual branch.  The reference always
    starts from ImageNet statistics (retinanet_cfg.py:8); with identity BN a random-init ResNet-50 overflows, so the
    benchmark's synthetic weights use a damped value (same FLOPs / bytes, finite activations)."""
    bu = "backbone.bottom_up"
    p[bu + ".conv1.weight"] = _msra_normal(rng, (64, 3, 7, 7), "fan_out")
    _bn(p, bu + ".bn1", 64)
    for blk in resnet_conv_table(name):
        pre = blk["prefix"]
        if blk["kind"] == "bottleneck":
            specs = [("conv1", blk["cin"], blk["ch"], 1), ("conv2", blk["ch"], blk["ch"], 3), ("conv3", blk["ch"], blk["cout"], 1)]
        else:
            specs = [("conv1", blk["cin"], blk["ch"], 3), ("conv2", blk["ch"], blk["cout"], 3)]
        for i, (cn, ci, co, k) in enumerate(specs):
            p[f"{pre}.{cn}.weight"] = _msra_normal(rng, (co, ci, k, k), "fan_out")
            _bn(p, f"{pre}.bn{i + 1}", co)
            if residual_gamma is not None and i + 1 == len(specs):
                p[f"{pre}.bn{i + 1}.weight"][:] = residual_gamma
        if blk["has_ds"]:
            p[f"{pre}.downsample.0.weight"] = _msra_normal(rng, (blk["cout"], blk["cin"], 1, 1), "fan_out")
            _bn(p, f"{pre}.downsample.1", blk["cout"])


def init_fpn(p, rng, in_channels, stages, out_ch, top_in, top_convs=True):
    for s, ci in zip(stages, in_channels):
        p[f"backbone.fpn_lateral{s}.weight"] = _msra_normal(rng, (out_ch, ci, 1, 1), "fan_in")
        p[f"backbone.fpn_lateral{s}.bias"] = np.zeros(out_ch, np.float32)
        p[f"backbone.fpn_output{s}.weight"] = _msra_normal(rng, (out_ch, out_ch, 3, 3), "fan_in")
        p[f"backbone.fpn_output{s}.bias"] = np.zeros(out_ch, np.float32)
    if not top_convs:      # FPNP6 (fpn_backbone.py:172-183) has no parameters
        return
    for nm, ci in (("p6", top_in), ("p7", out_ch)):
        p[f"backbone.top_block.{nm}.weight"] = (rng.standard_normal((out_ch, ci, 3, 3)) * math.sqrt(1.0 / (ci * 9))).astype(np.float32)
        p[f"backbone.top_block.{nm}.bias"] = np.zeros(out_ch, np.float32)


def init_retina_head(p, rng, ch, num_anchors, num_classes, num_convs, prior_prob):
    for tower in ("cls_subnet", "bbox_subnet"):
        for i in range(num_convs):
            p[f"head.{tower}.{2 * i}.weight"] = (rng.standard_normal((ch, ch, 3, 3)) * 0.01).astype(np.float32)
            p[f"head.{tower}.{2 * i}.bias"] = np.zeros(ch, np.float32)
    p["head.cls_score.weight"] = (rng.standard_normal((num_anchors * num_classes, ch, 3, 3)) * 0.01).astype(np.float32)
    p["head.cls_score.bias"] = np.full(num_anchors * num_classes, -math.log((1 - prior_prob) / prior_prob), np.float32)
    p["head.bbox_pred.weight"] = (rng.standard_normal((num_anchors * 4, ch, 3, 3)) * 0.01).astype(np.float32)
    p["head.bbox_pred.bias"] = np.zeros(num_anchors * 4, np.float32)


def init_retinanet_params(cfg, seed=0, residual_gamma=None):
    rng = np.random.default_rng(seed)
    m = cfg.MODEL
    p = {}
    init_resnet(p, rng, m.BACKBONE.NAME, residual_gamma)
    stages = [int(f[-1]) for f in m.BACKBONE.OUT_FEATURES]
    init_fpn(p, rng, m.BACKBONE.OUT_FEATURE_CHANNELS, stages, m.FPN.OUT_CHANNELS, m.FPN.TOP_BLOCK_IN_CHANNELS)
    na = len(m.ANCHOR.SCALES[0]) * len(m.ANCHOR.RATIOS[0])
    init_retina_head(p, rng, m.FPN.OUT_CHANNELS, na, cfg.DATA.NUM_CLASSES, m.HEAD.NUM_CONVS, m.HEAD.CLS_PRIOR_PROB)
    return p


def init_point_head(p, rng, ch, num_anchors, num_classes, num_convs, prior_prob, num_levels):
    """PointHead (layers/head/point_head.py:40-125): Sequential = [conv, GroupNorm(32), ReLU] x num_convs, so conv i sits at
    index 3i and its GroupNorm at 3i+1; N(0, 0.01) conv weights, zero bias, GN gamma=1 beta=0, scales = 1."""
    for tower in ("cls_subnet", "bbox_subnet"):
        for i in range(num_convs):
            p[f"head.{tower}.{3 * i}.weight"] = (rng.standard_normal((ch, ch, 3, 3)) * 0.01).astype(np.float32)
            p[f"head.{tower}.{3 * i}.bias"] = np.zeros(ch, np.float32)
            p[f"head.{tower}.{3 * i + 1}.weight"] = np.ones(ch, np.float32)
            p[f"head.{tower}.{3 * i + 1}.bias"] = np.zeros(ch, np.float32)
    p["head.cls_score.weight"] = (rng.standard_normal((num_anchors * num_classes, ch, 3, 3)) * 0.01).astype(np.float32)
    p["head.cls_score.bias"] = np.full(num_anchors * num_classes, -math.log((1 - prior_prob) / prior_prob), np.float32)
    p["head.bbox_pred.weight"] = (rng.standard_normal((num_anchors * 4, ch, 3, 3)) * 0.01).astype(np.float32)
    p["head.bbox_pred.bias"] = np.zeros(num_anchors * 4, np.float32)
    p["head.ctrness.weight"] = (rng.standard_normal((num_anchors, ch, 3, 3)) * 0.01).astype(np.float32)
    p["head.ctrness.bias"] = np.zeros(num_anchors, np.float32)
    p["head.scales"] = np.ones(num_levels, np.float32)


def init_fcos_params(cfg, seed=0, residual_gamma=None):
    rng = np.random.default_rng(seed)
    m = cfg.MODEL
    p = {}
    init_resnet(p, rng, m.BACKBONE.NAME, residual_gamma)
    stages = [int(f[-1]) for f in m.BACKBONE.OUT_FEATURES]
    init_fpn(p, rng, m.BACKBONE.OUT_FEATURE_CHANNELS, stages, m.FPN.OUT_CHANNELS, m.FPN.TOP_BLOCK_IN_CHANNELS)
    init_point_head(p, rng, m.FPN.OUT_CHANNELS, m.ANCHOR.NUM_ANCHORS, cfg.DATA.NUM_CLASSES, m.HEAD.NUM_CONVS,
                    m.HEAD.CLS_PRIOR_PROB, len(m.FPN.STRIDES))
    return p


def init_faster_rcnn_params(cfg, seed=0, residual_gamma=None):
    """FasterRCNN (models/det/faster_rcnn.py:19-44): ResNet + FPN(p2-p5, FPNP6) + RPN (rpn.py:52-68: N(0, 0.01), zero bias)
    + RCNN (layers/head/rcnn.py:32-50: fc1/fc2/pred_cls N(0, 0.01), pred_delta N(0, 0.001), zero bias)."""
    rng = np.random.default_rng(seed)
    m = cfg.MODEL
    p = {}
    init_resnet(p, rng, m.BACKBONE.NAME, residual_gamma)
    stages = [int(f[-1]) for f in m.BACKBONE.OUT_FEATURES]
    ch = m.FPN.OUT_CHANNELS
    init_fpn(p, rng, m.BACKBONE.OUT_FEATURE_CHANNELS, stages, ch, m.FPN.TOP_BLOCK_IN_CHANNELS, top_convs=False)
    na = len(m.ANCHOR.SCALES[0]) * len(m.ANCHOR.RATIOS[0])
    rc = m.RPN.CHANNELS
    p["rpn.rpn_conv.weight"] = (rng.standard_normal((rc, ch, 3, 3)) * 0.01).astype(np.float32)
    p["rpn.rpn_conv.bias"] = np.zeros(rc, np.float32)
    p["rpn.rpn_cls_score.weight"] = (rng.standard_normal((na, rc, 1, 1)) * 0.01).astype(np.float32)
    p["rpn.rpn_cls_score.bias"] = np.zeros(na, np.float32)
    p["rpn.rpn_bbox_offsets.weight"] = (rng.standard_normal((na * 4, rc, 1, 1)) * 0.01).astype(np.float32)
    p["rpn.rpn_bbox_offsets.bias"] = np.zeros(na * 4, np.float32)
    ph, pw = m.ROI_POOLER.SIZE
    K = cfg.DATA.NUM_CLASSES
    p["rcnn.fc1.weight"] = (rng.standard_normal((1024, ch * ph * pw)) * 0.01).astype(np.float32)
    p["rcnn.fc1.bias"] = np.zeros(1024, np.float32)
    p["rcnn.fc2.weight"] = (rng.standard_normal((1024, 1024)) * 0.01).astype(np.float32)
    p["rcnn.fc2.bias"] = np.zeros(1024, np.float32)
    p["rcnn.pred_cls.weight"] = (rng.standard_normal((K + 1, 1024)) * 0.01).astype(np.float32)
    p["rcnn.pred_cls.bias"] = np.zeros(K + 1, np.float32)
    p["rcnn.pred_delta.weight"] = (rng.standard_normal((K * 4, 1024)) * 0.001).astype(np.float32)
    p["rcnn.pred_delta.bias"] = np.zeros(K * 4, np.float32)
    return p


def trainable_names(params, freeze_at=2):
    """DetSolver.params (solver/default_solver.py:83-94): drop bottom_up.conv1 / layer1 by name; FrozenBN
    statistics and affine terms never receive gradients (configs/extra_cfg.py:55)."""
    out = []
    for k in params:
        if ".bn" in k or "downsample.1" in k or "running_" in k:
            continue
        if "bottom_up.conv1" in k and freeze_at >= 1:
            continue
        if "bottom_up.layer1" in k and freeze_at >= 2:
            continue
        out.append(k)
    return out


def oracle_arch(cfg):
    m = cfg.MODEL
    if m.NAME == "FasterRCNN":
        return dict(backbone=m.BACKBONE.NAME, fpn_in=list(m.BACKBONE.OUT_FEATURES), top_block="pool",
                    num_classes=cfg.DATA.NUM_CLASSES, img_mean=list(m.BACKBONE.IMG_MEAN), img_std=list(m.BACKBONE.IMG_STD),
                    strides=list(m.FPN.STRIDES), anchor_scales=[list(s) for s in m.ANCHOR.SCALES],
                    anchor_ratios=[list(r) for r in m.ANCHOR.RATIOS], anchor_offset=m.ANCHOR.OFFSET,
                    rpn=dict(m.RPN), rcnn=dict(m.RCNN), pool_size=tuple(m.ROI_POOLER.SIZE),
                    rpn_box_reg=(list(m.RPN_BOX_REG.MEAN), list(m.RPN_BOX_REG.STD)),
                    rcnn_box_reg=(list(m.RCNN_BOX_REG.MEAN), list(m.RCNN_BOX_REG.STD)),
                    matcher=(list(m.MATCHER.THRESHOLDS), list(m.MATCHER.LABELS), m.MATCHER.ALLOW_LOW_QUALITY),
                    rpn_beta=m.LOSSES.RPN_SMOOTH_L1_BETA, rcnn_beta=m.LOSSES.RCNN_SMOOTH_L1_BETA)
    if m.NAME in ("FCOS", "ATSS", "OTA"):
        extra = dict(atss=dict(scale=m.ANCHOR.SCALE, topk=m.ANCHOR.TOPK), sizes_of_interest=None, center_sampling_radius=None) \
            if m.NAME == "ATSS" else dict(sizes_of_interest=[list(s) for s in m.HEAD.OBJECT_SIZES_OF_INTEREST],
                                          center_sampling_radius=m.HEAD.CENTER_SAMPLING_RADIUS)
        if m.NAME == "OTA":
            extra["ota"] = dict(reg_weight=m.HEAD.get("COST_REG_WEIGHTS", 1.5), candidate_k=m.HEAD.get("CANDIDATE_K", 10),
                                center_radius=2.5, matching=m.get("MATCHING", "topk"))
        return dict(extra, backbone=m.BACKBONE.NAME, fpn_in=list(m.BACKBONE.OUT_FEATURES), num_convs=m.HEAD.NUM_CONVS,
                    num_classes=cfg.DATA.NUM_CLASSES, img_mean=list(m.BACKBONE.IMG_MEAN), img_std=list(m.BACKBONE.IMG_STD),
                    strides=list(m.FPN.STRIDES), anchor_offset=m.ANCHOR.OFFSET,
                    focal_alpha=m.LOSSES.FOCAL_LOSS_ALPHA, focal_gamma=m.LOSSES.FOCAL_LOSS_GAMMA,
                    iou_loss_type=m.LOSSES.IOU_LOSS_TYPE, reg_loss_weight=m.LOSSES.REG_LOSS_WEIGHT)
    extra = dict(freeanchor=dict(mean=list(m.BOX_REG.MEAN), std=list(m.BOX_REG.STD), iou_thresh=m.BUCKET.BOX_IOU_THRESH,
                                 bucket=m.BUCKET.BUCKET_SIZE)) if m.NAME == "FreeAnchor" else {}
    return dict(extra, backbone=m.BACKBONE.NAME, fpn_in=list(m.BACKBONE.OUT_FEATURES), num_convs=m.HEAD.NUM_CONVS,
                num_classes=cfg.DATA.NUM_CLASSES, img_mean=list(m.BACKBONE.IMG_MEAN), img_std=list(m.BACKBONE.IMG_STD),
                strides=list(m.FPN.STRIDES), anchor_scales=[list(s) for s in m.ANCHOR.SCALES],
                anchor_ratios=[list(r) for r in m.ANCHOR.RATIOS], anchor_offset=m.ANCHOR.OFFSET,
                focal_alpha=m.LOSSES.FOCAL_LOSS_ALPHA, focal_gamma=m.LOSSES.FOCAL_LOSS_GAMMA,
                smooth_l1_beta=m.LOSSES.SMOOTH_L1_BETA, reg_loss_weight=m.LOSSES.REG_LOSS_WEIGHT)
