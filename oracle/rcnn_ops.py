"""CPU ORACLE (test infrastructure only) -- numpy restatement of the reference's Faster R-CNN operators.

This file is NOT part of the product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Every function cites the reference file:line it restates
(paths relative to the reference repo ``basedet/``).

PARITY: RoIAlign and max RoI pooling are PINNED by the reference's own known-answer tests (tests/layers/test_roi_pool.py:32-61: the
two 4x4 matrices on the 5x5 arange map; :64-75: invariance under a 2x bilinear upsampling at stride 1/2), carried as data in
tests/golden/reference_kat.npz and asserted on this file by tests/test_oracle_rcnn_cpu.py (and on the HIP kernels by
tests/test_rcnn_ops_gpu.py / tests/test_layers_gpu.py).  PARITY UNPINNED for the rest (RPN / RCNN target sampling, proposal
selection, the losses): the reference holds no vector for them and cannot be imported here (megengine / basecore are absent).
Two pieces live inside MegEngine itself (an un-vendored dependency, ``megengine>=1.8`` in the reference's requirements) and are
restated from their published algorithms:
  * ``F.nn.roi_align(mode="average", sample_points=2, aligned=True)`` -- the Detectron / caffe2 RoIAlign the MegEngine
    kernel is derived from (bilinear_interpolate with the [-1, size] validity window, continuous coordinates shifted
    by -0.5 when aligned, no minimum RoI size when aligned); ``F.nn.roi_pooling(mode="max")`` -- Caffe's ROIPooling (rounded
    corners, inclusive extent, floor / ceil bin edges);
  * ``F.topk`` -- k > 0 smallest / k < 0 largest, so ``sample_labels`` (sampling.py:27) with its negative k marks the
    entries with the LARGEST random keys as ignored.
Documented choices where the reference is silent or random:
  * top-k / sort tie-break: score descending, then index ascending; "score" order is the total order of the float bit
    patterns (so -0.0 < +0.0), matching the radix keys of the HIP kernels;
  * the random keys of ``sample_labels`` are an INPUT (the reference draws them from megengine.random.uniform):
    with equal keys the lower index survives;
  * proposals are the CLIPPED boxes (rpn.py:170 clips a ``Boxes`` view in place; detectron2 semantics);
  * an image without ground truth: every RoI is background with zero targets (the reference would raise).
"""
import math

import numpy as np

from . import box_ops as B

F32 = np.float32


def _asc_key(v):
    """Total order on float32 bit patterns (the radix key of csrc/rcnn_ops.hip f32_asc_key)."""
    u = np.asarray(v, F32).view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000).astype(np.uint64)


def topk_desc(scores, k, min_score=None):
    """F.topk(scores, k, descending=True) (rpn.py:155) -> (indices, scores), stable."""
    scores = np.asarray(scores, F32)
    idx = np.arange(scores.shape[0])
    if min_score is not None:
        idx = idx[scores > F32(min_score)]
    key = _asc_key(scores[idx])
    order = np.lexsort((idx, -key.astype(np.int64)))
    sel = idx[order][:k]
    return sel.astype(np.int32), scores[sel]


def sample_labels(labels, keys, num_samples, label_value, ignore_label=-1):
    """sampling.py:7-30 with the random tensor supplied: the (num_valid - num_samples) entries with the largest keys
    among ``labels == label_value`` become ``ignore_label``."""
    labels = np.asarray(labels).copy()
    mask = labels == label_value
    num_valid = int(mask.sum())
    if num_valid <= num_samples:
        return labels
    idx = np.nonzero(mask)[0]
    order = np.lexsort((idx, np.asarray(keys, F32)[idx]))     # keys ascending, then index ascending
    drop = idx[order][max(num_samples, 0):]
    labels[drop] = ignore_label
    return labels


def rpn_ground_truth(anchors, batched_gt_boxes, num_valid, keys_pos, keys_neg, thresholds=(0.3, 0.7), labels=(0, -1, 1),
                     allow_low_quality=True, num_sample_anchors=256, num_pos_anchor=128, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    """RPN.get_ground_truth (rpn.py:215-240).  Returns labels (N, A) in {-1, 0, 1}, offsets (N, A, 4)."""
    labs, offs = [], []
    for bid, (gtb, n) in enumerate(zip(batched_gt_boxes, num_valid)):
        gt = np.asarray(gtb, F32)[: int(n)]
        A = len(anchors)
        if gt.shape[0] == 0:
            lab = np.zeros(A, np.int32)
            off = np.zeros((A, 4), F32)
        else:
            overlaps = B.box_iou(gt[:, :4], anchors)
            idx, lab = B.matcher(overlaps, list(thresholds), list(labels), allow_low_quality)
            off = B.box_encode(anchors, gt[idx, :4], mean, std)
        lab = sample_labels(lab, keys_pos[bid], num_pos_anchor, 1, -1)
        num_negative = num_sample_anchors - int((lab == 1).sum())
        lab = sample_labels(lab, keys_neg[bid], num_negative, 0, -1)
        labs.append(lab.astype(np.int32))
        offs.append(off)
    return np.stack(labs), np.stack(offs)


def rpn_proposals(scores_per_level, offsets_per_level, anchors_per_level, im_hw, pre_k, post_k, nms_thresh,
                  mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    """RPN.find_top_rpn_proposals (rpn.py:134-186) for ONE image.  scores_per_level[l]: (H*W*A,), offsets (H*W*A, 4),
    anchors (H*W*A, 4).  Returns (rois (n, 4), scores (n,), candidates dict for debugging)."""
    props, scs, lvls = [], [], []
    for level, (s, o, a) in enumerate(zip(scores_per_level, offsets_per_level, anchors_per_level)):
        order, top = topk_desc(s, pre_k)
        p = B.box_decode(np.asarray(a, F32)[order], np.asarray(o, F32)[order], mean, std)
        props.append(p); scs.append(top); lvls.append(np.full(len(order), level, np.int32))
    props = np.concatenate(props); scs = np.concatenate(scs); lvls = np.concatenate(lvls)
    clipped = B.box_clip(props, im_hw)
    keep_mask = ((clipped[:, 2] - clipped[:, 0]) > 0) & ((clipped[:, 3] - clipped[:, 1]) > 0)     # boxes.py:132-150
    kept_boxes, kept_scores, kept_lvls = clipped[keep_mask], scs[keep_mask], lvls[keep_mask]
    keep = B.batched_nms(kept_boxes, kept_scores, kept_lvls, nms_thresh, post_k)
    return kept_boxes[keep], kept_scores[keep], dict(boxes=clipped, scores=scs, levels=lvls, valid=keep_mask)


def rcnn_ground_truth(rois, gt_boxes_with_labels, keys_fg, keys_bg, num_rois=512, fg_ratio=0.5, fg_thresh=0.5,
                      bg_thresh_high=0.5, bg_thresh_low=0.0, mean=(0, 0, 0, 0), std=(0.1, 0.1, 0.2, 0.2)):
    """RCNN.get_ground_truth (rcnn.py:95-147) for ONE image.  rois (n, 4) proposals, gt (G, 5).
    Returns kept rois (m, 4), labels (m,) int32, bbox_targets (m, 4)."""
    gt = np.asarray(gt_boxes_with_labels, F32)
    all_rois = np.concatenate([np.asarray(rois, F32).reshape(-1, 4), gt[:, :4]], 0)
    M = all_rois.shape[0]
    if gt.shape[0] == 0:
        max_ov = np.zeros(M, F32); assign = np.zeros(M, np.int32); labels = np.zeros(M, F32)
    else:
        overlaps = B.box_iou(all_rois, gt[:, :4])
        max_ov = overlaps.max(axis=1)
        assign = overlaps.argmax(axis=1).astype(np.int32)
        labels = gt[assign, 4].copy()
    fg_mask = (max_ov >= F32(fg_thresh)) & (labels >= 0) & (gt.shape[0] > 0)
    bg_mask = (max_ov >= F32(bg_thresh_low)) & (max_ov < F32(bg_thresh_high))
    num_fg_rois = int(num_rois * fg_ratio)
    fg_inds = sample_labels(fg_mask, keys_fg[:M], num_fg_rois, True, False)
    num_bg_rois = int(num_rois - fg_inds.sum())
    bg_inds = sample_labels(bg_mask, keys_bg[:M], num_bg_rois, True, False)
    labels[bg_inds] = 0
    keep = fg_inds | bg_inds
    out_rois = all_rois[keep]
    out_labels = labels[keep].astype(np.int32)
    if gt.shape[0] == 0:
        targets = np.zeros((out_rois.shape[0], 4), F32)
    else:
        targets = B.box_encode(out_rois, gt[assign[keep], :4], mean, std)
    return out_rois, out_labels, targets


def assign_roi_levels(rois, strides):
    """assign_rois (roi_pool.py:12-25) without the dummy rows: level index into ``strides``."""
    rois = np.asarray(rois, F32)
    min_level, max_level = int(math.log2(strides[0])), int(math.log2(strides[-1]))
    area = (rois[:, 2] - rois[:, 0]) * (rois[:, 3] - rois[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        v = F32(4) + np.log(np.sqrt(area).astype(F32) / F32(224)).astype(F32) / F32(math.log(2))
    lvl = np.full(rois.shape[0], min_level, np.int64)
    ok = np.isfinite(v)
    lvl[ok] = np.floor(v[ok]).astype(np.int64)
    pos_inf = np.isposinf(v)
    lvl[pos_inf] = max_level
    lvl = np.clip(lvl, min_level, max_level) - min_level
    return lvl.astype(np.int32)


def _bilinear(feat, y, x):
    """caffe2 / Detectron bilinear_interpolate on feat (H, W, C) at scalar (y, x), float32."""
    H, W = feat.shape[0], feat.shape[1]
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return None
    y = F32(max(y, 0.0)); x = F32(max(x, 0.0))
    y0, x0 = int(y), int(x)
    if y0 >= H - 1:
        y0 = y1 = H - 1; y = F32(y0)
    else:
        y1 = y0 + 1
    if x0 >= W - 1:
        x0 = x1 = W - 1; x = F32(x0)
    else:
        x1 = x0 + 1
    ly = F32(y - F32(y0)); lx = F32(x - F32(x0)); hy = F32(1) - ly; hx = F32(1) - lx
    return (y0, y1, x0, x1, F32(hy * hx), F32(hy * lx), F32(ly * hx), F32(ly * lx))


def roi_align_sample_table(roi, scale, H, W, PH=7, PW=7, S=2):
    """Sample points of one RoI: list over bins of [(y0, y1, x0, x1, w00, w01, w10, w11), ...] (None = outside)."""
    x1, y1, x2, y2 = [F32(v) for v in roi]
    sw = F32(x1 * F32(scale) - F32(0.5)); sh = F32(y1 * F32(scale) - F32(0.5))
    rw = F32(F32(x2 * F32(scale) - F32(0.5)) - sw); rh = F32(F32(y2 * F32(scale) - F32(0.5)) - sh)
    bw = F32(rw / F32(PW)); bh = F32(rh / F32(PH))
    table = []
    for ph in range(PH):
        for pw in range(PW):
            pts = []
            for iy in range(S):
                y = F32(F32(sh + F32(F32(ph) * bh)) + F32(F32(F32(iy) + F32(0.5)) * bh) / F32(S))
                for ix in range(S):
                    x = F32(F32(sw + F32(F32(pw) * bw)) + F32(F32(F32(ix) + F32(0.5)) * bw) / F32(S))
                    pts.append(_bilinear(np.empty((H, W, 0)), float(y), float(x)))
            table.append(pts)
    return table


def roi_align(feats, rois, batch_idx, strides, PH=7, PW=7, S=2):
    """roi_pool(..., "roi_align") (roi_pool.py:35-78).  feats[l]: (N, H, W, C) float32 (channel-last), rois (R, 4),
    batch_idx (R,).  Returns (R, PH*PW, C) float32 (bin-major)."""
    lv = assign_roi_levels(rois, strides)
    C = feats[0].shape[-1]
    out = np.zeros((len(rois), PH * PW, C), F32)
    for r, (roi, n, l) in enumerate(zip(np.asarray(rois, F32), batch_idx, lv)):
        f = feats[l][int(n)]
        table = roi_align_sample_table(roi, 1.0 / strides[l], f.shape[0], f.shape[1], PH, PW, S)
        for b, pts in enumerate(table):
            acc = np.zeros(C, F32)
            for p in pts:
                if p is None:
                    continue
                y0, y1, x0, x1, w00, w01, w10, w11 = p
                acc += w00 * f[y0, x0] + w01 * f[y0, x1] + w10 * f[y1, x0] + w11 * f[y1, x1]
            out[r, b] = acc * F32(1.0 / (S * S))
    return out


def roi_pool_max(feat, rois5, scale, PH, PW):
    """roi_pool(..., "roi_pool") = F.nn.roi_pooling(mode="max") (roi_pool.py:65) on one level: Caffe ROIPooling.
    feat (N, C, H, W), rois5 (R, 5) = (batch index, x1, y1, x2, y2) -> (R, C, PH, PW); an empty bin gives 0."""
    feat = np.asarray(feat, F32)
    _, C, H, W = feat.shape
    out = np.zeros((len(rois5), C, PH, PW), F32)
    for r, roi in enumerate(np.asarray(rois5, F32)):
        n = int(roi[0])
        x1, y1, x2, y2 = [int(np.floor(F32(v) * F32(scale) + F32(0.5))) for v in roi[1:]]      # C roundf for non-negative inputs
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        bh, bw = F32(rh) / F32(PH), F32(rw) / F32(PW)
        for ph in range(PH):
            hs = min(max(int(np.floor(F32(ph) * bh)) + y1, 0), H); he = min(max(int(np.ceil(F32(ph + 1) * bh)) + y1, 0), H)
            for pw in range(PW):
                ws = min(max(int(np.floor(F32(pw) * bw)) + x1, 0), W); we = min(max(int(np.ceil(F32(pw + 1) * bw)) + x1, 0), W)
                if he > hs and we > ws:
                    out[r, :, ph, pw] = feat[n, :, hs:he, ws:we].max(axis=(1, 2))
    return out


def roi_align_backward(gout, feats_shapes, rois, batch_idx, strides, PH=7, PW=7, S=2):
    """Adjoint of roi_align: gout (R, PH*PW, C) -> list of (N, H, W, C) float64-accumulated gradients."""
    lv = assign_roi_levels(rois, strides)
    grads = [np.zeros(s, np.float64) for s in feats_shapes]
    for r, (roi, n, l) in enumerate(zip(np.asarray(rois, F32), batch_idx, lv)):
        g = grads[l][int(n)]
        table = roi_align_sample_table(roi, 1.0 / strides[l], g.shape[0], g.shape[1], PH, PW, S)
        for b, pts in enumerate(table):
            go = gout[r, b].astype(np.float64) / (S * S)
            for p in pts:
                if p is None:
                    continue
                y0, y1, x0, x1, w00, w01, w10, w11 = p
                g[y0, x0] += w00 * go; g[y0, x1] += w01 * go; g[y1, x0] += w10 * go; g[y1, x1] += w11 * go
    return grads


def rpn_losses(logits, offsets, labels, targets, beta=0.0):
    """rpn.py:113-131 on flat arrays: logits (M,), offsets (M, 4), labels (M,), targets (M, 4) -> (cls, bbox)."""
    logits = np.asarray(logits, np.float64); labels = np.asarray(labels)
    valid = labels >= 0
    fg = labels > 0
    num_valid = int(valid.sum())
    x, t = logits[valid], labels[valid].astype(np.float64)
    ls = lambda v: np.minimum(v, 0) - np.log1p(np.exp(-np.abs(v)))
    cls = float((-(t * ls(x) + (1 - t) * ls(-x))).mean()) if num_valid else 0.0
    d = np.asarray(offsets, np.float64)[fg] - np.asarray(targets, np.float64)[fg]
    if beta < 1e-5:
        l = np.abs(d)
    else:
        l = np.where(np.abs(d) < beta, 0.5 * d * d / beta, np.abs(d) - 0.5 * beta)
    return cls, float(l.sum() / max(num_valid, 1))


def rcnn_losses(logits, deltas, labels, targets, beta=0.0):
    """rcnn.py:65-83: logits (R, K+1), deltas (R, K, 4), labels (R,) >= 0, targets (R, 4)."""
    logits = np.asarray(logits, np.float64)
    labels = np.asarray(labels)
    R = logits.shape[0]
    m = logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(logits - m).sum(axis=1)) + m[:, 0]
    cls = float((lse - logits[np.arange(R), labels]).mean()) if R else 0.0
    fg = labels > 0
    d = np.asarray(deltas, np.float64)[fg, labels[fg] - 1] - np.asarray(targets, np.float64)[fg]
    if beta < 1e-5:
        l = np.abs(d)
    else:
        l = np.where(np.abs(d) < beta, 0.5 * d * d / beta, np.abs(d) - 0.5 * beta)
    return cls, float(l.sum() / max(R, 1))


# --------------------------------------------------------------------------------------------
# inference post-processing (models/det/retinanet.py:172-209, fcos.py:181-216, layers/common/post_processing.py:50-103)
# --------------------------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x, F32)
    return (F32(1) / (F32(1) + np.exp(-x).astype(F32))).astype(F32)


def detect_postprocess(scores_per_level, boxes_per_level, num_classes, im_info, cls_threshold=0.05, iou_threshold=0.5,
                       max_detections=100, topk=1000):
    """scores_per_level[l]: flat (rows_l * K,) scores; boxes_per_level[l]: (rows_l, 4) decoded boxes -- or (rows_l * K, 4)
    per-item boxes (RCNN).  Returns (boxes, scores, labels) after NMS, rescaled to the original image and clipped."""
    tb, ts, tl = [], [], []
    for sc, bx in zip(scores_per_level, boxes_per_level):
        idx, top = topk_desc(sc, topk, cls_threshold)
        if len(idx) == 0:
            continue
        ts.append(top)
        tl.append((idx % num_classes).astype(np.int32))
        bx = np.asarray(bx, F32)
        tb.append(bx[idx] if bx.shape[0] == len(sc) else bx[idx // num_classes])
    if not tb:
        return np.zeros((0, 4), F32), np.zeros((0,), F32), np.zeros((0,), np.int32)
    boxes = np.concatenate(tb); scores = np.concatenate(ts); labels = np.concatenate(tl)
    keep = B.batched_nms(boxes, scores, labels, iou_threshold, max_detections)
    info = np.asarray(im_info, F32).reshape(-1)
    out = B.box_scale(boxes[keep], (F32(info[2] / info[0]), F32(info[3] / info[1])))
    out = B.box_clip(out, info[2:4])
    return out, scores[keep], labels[keep]
