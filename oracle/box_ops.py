"""CPU ORACLE (test infrastructure only) -- numpy restatement of the reference's box operators.

This file is NOT part of the product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Every function cites the reference file:line
it restates (paths relative to the reference repo ``basedet/``).

Pinning status:
  * pinned by the reference's own known-answer tests: ``box_iou`` / ``box_ioa`` / ``intersection`` /
    ``centers`` / ``scale`` (tests/structures/test_boxes.py:38-86), ``batched_nms``
    (tests/layers/test_postprocess.py:13-28), ``get_padded_tensor`` (tests/layers/test_preprocess.py:13-35).
  * PARITY UNPINNED (no reference test holds a value; the reference cannot be imported here because
    megengine/basecore are absent): anchors, Matcher, BoxCoder, RetinaNet/FCOS target assignment, losses.
    For those the restatement + the tie-break rules documented below are the only oracle.

All arithmetic is float32 with the reference's operation order so that integer outputs
(labels, matched indices, NMS keep lists) can be compared bit-exactly with the HIP kernels.
Tie-breaks (unpinned by the reference): argmax/argmin return the LOWEST index.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------
# anchors  (layers/common/anchor_generator.py)
# --------------------------------------------------------------------------------------------
def generate_base_anchors(scales, ratios):
    """anchor_generator.py:99-109 -- python float64 math, scale-major / ratio-minor, cast to f32 (:95)."""
    base = []
    for s in scales:
        area = float(s) ** 2.0
        for r in ratios:
            w = math.sqrt(area / float(r))
            h = float(r) * w
            base.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return np.asarray(base, dtype=F32)


def _listify_levels(v, n):
    v = [list(x) for x in v]
    if len(v) == 1:
        v = v * n
    assert len(v) == n
    return v


def create_anchor_grid(featmap_size, offset, stride):
    """anchor_generator.py:23-30 (+ function.py:47-54 meshgrid): row-major, x fastest."""
    h, w = featmap_size
    shift = F32(offset * stride)
    gx = (np.arange(w, dtype=F32) * F32(stride) + shift).astype(F32)
    gy = (np.arange(h, dtype=F32) * F32(stride) + shift).astype(F32)
    xs = np.broadcast_to(gx[None, :], (h, w)).reshape(-1)
    ys = np.broadcast_to(gy[:, None], (h, w)).reshape(-1)
    return xs, ys


def default_anchors(feat_sizes, strides, scales, ratios, offset):
    """DefaultAnchorGenerator.generate_anchors_by_features (anchor_generator.py:111-122)."""
    n = len(strides)
    # the reference stores scales/ratios as float32 arrays and converts back with tolist() (:73-83)
    scales = _listify_levels(np.asarray(scales, dtype=F32).tolist(), n)
    ratios = _listify_levels(np.asarray(ratios, dtype=F32).tolist(), n)
    out = []
    for size, stride, sc, ra in zip(feat_sizes, strides, scales, ratios):
        base = generate_base_anchors(sc, ra)
        xs, ys = create_anchor_grid(size, offset, stride)
        grids = np.stack([xs, ys, xs, ys], axis=1)
        out.append((grids[:, None, :] + base[None, :, :]).reshape(-1, 4).astype(F32))
    return out


def point_anchors(feat_sizes, strides, offset=0.5, num_anchors=1):
    """AnchorPointGenerator.generate_anchors_by_features (anchor_generator.py:152-165)."""
    out = []
    for size, stride in zip(feat_sizes, strides):
        xs, ys = create_anchor_grid(size, offset, stride)
        g = np.stack([xs, ys], axis=1)
        out.append(np.repeat(g[:, None, :], num_anchors, axis=1).reshape(-1, 2).astype(F32))
    return out


# --------------------------------------------------------------------------------------------
# pairwise box ops  (structures/op_patch.py, structures/boxes.py)
# --------------------------------------------------------------------------------------------
def intersection(b1, b2):
    """Boxes.intersection (boxes.py:114-130)."""
    b1 = np.asarray(b1, F32)[:, None, :]
    b2 = np.asarray(b2, F32)[None, :, :]
    iw = np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0])
    ih = np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1])
    return (np.maximum(iw, F32(0)) * np.maximum(ih, F32(0))).astype(F32)


def box_area(b):
    b = np.asarray(b, F32)
    return ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(F32)


def box_iou(b1, b2):
    """op_patch.py:33-97 IOU subgraph: inter / (a1 + a2 - inter), then max(., 0); no +1, no eps.
    0/0 (NaN) maps to 0 here (fmax semantics), which is what the HIP kernel does too."""
    inter = intersection(b1, b2)
    union = (box_area(b1)[:, None] + box_area(b2)[None, :]).astype(F32) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter / union).astype(F32)
    return np.fmax(iou, F32(0)).astype(F32)


def box_ioa(b1, b2):
    """op_patch.py:170-227 IOA subgraph: inter / area(boxes2), max(., 0).  Result (len(b1), len(b2))."""
    inter = intersection(b1, b2)
    with np.errstate(divide="ignore", invalid="ignore"):
        ioa = (inter / box_area(b2)[None, :]).astype(F32)
    return np.fmax(ioa, F32(0)).astype(F32)


def box_giou(b1, b2):
    """Boxes.giou (boxes.py:74-95)."""
    b1 = np.asarray(b1, F32)
    b2 = np.asarray(b2, F32)
    inter = intersection(b1, b2)
    union = (box_area(b1)[:, None] + box_area(b2)[None, :]).astype(F32) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
        x1, x2 = b1[:, None, :], b2[None, :, :]
        lt = np.minimum(x1[..., :2], x2[..., :2])
        rb = np.maximum(x1[..., 2:], x2[..., 2:])
        wh = np.maximum(rb - lt, F32(0))
        area = wh[..., 0] * wh[..., 1]
        return (iou - (area - union) / area).astype(F32)


def box_centers(b):
    """op_patch.py:101-130: (top_left + bottom_right) / 2."""
    b = np.asarray(b, F32)
    return ((b[:, :2] + b[:, 2:]) / F32(2)).astype(F32)


def box_scale(b, ratios):
    """Boxes.scale (boxes.py:193-...): ratios = (h_ratio, w_ratio) or scalar."""
    b = np.asarray(b, F32).copy()
    if isinstance(ratios, (int, float)):
        ratios = (ratios, ratios)
    rh, rw = F32(ratios[0]), F32(ratios[1])
    b[:, 0::2] *= rw
    b[:, 1::2] *= rh
    return b


def box_clip(b, sizes):
    """Boxes.clip (boxes.py:152-178): sizes = (h, w)."""
    b = np.asarray(b, F32).copy()
    h, w = F32(sizes[0]), F32(sizes[1])
    b[:, 0::2] = np.clip(b[:, 0::2], F32(0), w)
    b[:, 1::2] = np.clip(b[:, 1::2], F32(0), h)
    return b


# --------------------------------------------------------------------------------------------
# Matcher (layers/common/matcher.py:19-51)
# --------------------------------------------------------------------------------------------
def matcher(matrix, thresholds, labels, allow_low_quality=False):
    """matrix (G, A).  Half-open bands ``low <= v < high`` (:44); low-quality rule (:47-49) does not
    change match_indices.  argmax tie-break: lowest gt index."""
    matrix = np.asarray(matrix, F32)
    thr = [-float("inf")] + list(thresholds) + [float("inf")]
    max_scores = matrix.max(axis=0)
    idx = matrix.argmax(axis=0).astype(np.int32)
    out = np.full(idx.shape, -1, dtype=np.int32)
    for lab, lo, hi in zip(labels, thr[:-1], thr[1:]):
        out[(max_scores >= lo) & (max_scores < hi)] = lab
    if allow_low_quality:
        m = (matrix == matrix.max(axis=1, keepdims=True)).sum(axis=0) > 0
        out[m] = 1
    return idx, out


# --------------------------------------------------------------------------------------------
# BoxCoder / PointCoder (structures/boxcoder.py)
# --------------------------------------------------------------------------------------------
def _ltrb_to_cs(b):
    w = b[:, 2] - b[:, 0]
    h = b[:, 3] - b[:, 1]
    return w, h, b[:, 0] + F32(0.5) * w, b[:, 1] + F32(0.5) * h


def box_encode(anchors, gt, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    """BoxCoder.encode (boxcoder.py:61-73)."""
    anchors = np.asarray(anchors, F32)
    gt = np.asarray(gt, F32)
    aw, ah, acx, acy = _ltrb_to_cs(anchors)
    gw, gh, gcx, gcy = _ltrb_to_cs(gt)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.stack([(gcx - acx) / aw, (gcy - acy) / ah, np.log(gw / aw), np.log(gh / ah)], axis=1)
    t = t.astype(F32)
    t = (t - np.asarray(mean, F32)[None]) / np.asarray(std, F32)[None]
    return t.astype(F32)


def box_decode(anchors, deltas, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    """BoxCoder.decode (boxcoder.py:75-98).  deltas (A, 4*k)."""
    anchors = np.asarray(anchors, F32)
    d = np.asarray(deltas, F32)
    k = d.shape[1] // 4
    d = d * np.tile(np.asarray(std, F32), k)[None] + np.tile(np.asarray(mean, F32), k)[None]
    aw, ah, acx, acy = [v[:, None] for v in _ltrb_to_cs(anchors)]
    cx = acx + d[:, 0::4] * aw
    cy = acy + d[:, 1::4] * ah
    w = aw * np.exp(d[:, 2::4])
    h = ah * np.exp(d[:, 3::4])
    out = np.stack([cx - F32(0.5) * w, cy - F32(0.5) * h, cx + F32(0.5) * w, cy + F32(0.5) * h], axis=2)
    return out.reshape(out.shape[0], -1).astype(F32)


def point_encode(points, gt):
    """PointCoder.encode (boxcoder.py:132-133): (px-x1, py-y1, x2-px, y2-py); broadcasts."""
    points = np.asarray(points, F32)
    gt = np.asarray(gt, F32)
    return np.concatenate([points - gt[..., :2], gt[..., 2:] - points], axis=-1).astype(F32)


def point_decode(points, deltas):
    """PointCoder.decode (boxcoder.py:135-141)."""
    p = np.asarray(points, F32)
    d = np.asarray(deltas, F32)
    out = np.stack([p[:, 0:1] - d[:, 0::4], p[:, 1:2] - d[:, 1::4], p[:, 0:1] + d[:, 2::4], p[:, 1:2] + d[:, 3::4]], axis=2)
    return out.reshape(d.shape).astype(F32)


# --------------------------------------------------------------------------------------------
# RetinaNet target assignment (models/det/retinanet.py:211-232)
# --------------------------------------------------------------------------------------------
def retinanet_ground_truth(anchors, batched_gt_boxes, num_valid, thresholds=(0.4, 0.5), labels=(0, -1, 1),
                           allow_low_quality=True, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    """Returns labels (N, A) int32 {-1 ignore, 0 bg, k>0 class}, offsets (N, A, 4) f32, match idx (N, A)."""
    labs, offs, idxs = [], [], []
    for boxes_with_labels, n in zip(batched_gt_boxes, num_valid):
        gt = np.asarray(boxes_with_labels, F32)[: int(n)]
        if gt.shape[0] == 0:   # the reference would fail on an empty max(); defined here as "all background"
            A = len(anchors)
            labs.append(np.zeros(A, np.int32)); offs.append(np.zeros((A, 4), F32)); idxs.append(np.zeros(A, np.int32))
            continue
        overlaps = box_iou(gt[:, :4], anchors)
        idx, lab = matcher(overlaps, list(thresholds), list(labels), allow_low_quality)
        matched = gt[idx]
        fg = lab == 1
        lab = lab.copy()
        lab[fg] = matched[fg, 4].astype(np.int32)
        off = box_encode(anchors, matched[:, :4], mean, std)
        labs.append(lab)
        offs.append(off)
        idxs.append(idx)
    return np.stack(labs), np.stack(offs), np.stack(idxs)


# --------------------------------------------------------------------------------------------
# FCOS target assignment (models/det/fcos.py:222-293)
# --------------------------------------------------------------------------------------------
def fcos_ground_truth(points_list, strides, batched_gt_boxes, num_valid, sizes_of_interest, radius=1.5):
    """Returns labels (N, P) int32, ltrb offsets (N, P, 4) f32, centerness (N, P) f32.
    argmin tie-break: lowest gt index."""
    all_pts = np.concatenate(points_list, axis=0).astype(F32)
    soi = np.concatenate([np.broadcast_to(np.asarray(s, F32)[None], (p.shape[0], 2)) for p, s in zip(points_list, sizes_of_interest)], 0)
    labs, offs, ctrs = [], [], []
    for boxes_with_labels, n in zip(batched_gt_boxes, num_valid):
        gtl = np.asarray(boxes_with_labels, F32)[: int(n)]
        gt = gtl[:, :4]
        offsets = point_encode(all_pts[None, :, :], gt[:, None, :])  # (G, P, 4)
        mx = offsets.max(axis=2)
        cared = (mx >= soi[None, :, 0]) & (mx <= soi[None, :, 1])
        if radius > 0:
            ctr = box_centers(gt)
            inb = []
            for stride, pts in zip(strides, points_list):
                r = F32(stride * radius)
                cb = np.concatenate([np.maximum(ctr - r, gt[:, :2]), np.minimum(ctr + r, gt[:, 2:4])], axis=1)
                co = point_encode(pts[None, :, :], cb[:, None, :])
                inb.append(co.min(axis=2) > 0)
            inb = np.concatenate(inb, axis=1)
        else:
            inb = offsets.min(axis=2) > 0
        areas = np.broadcast_to(box_area(gt)[:, None], offsets.shape[:2]).copy()
        areas[~cared] = np.inf
        areas[~inb] = np.inf
        if areas.shape[0] == 0:
            P = all_pts.shape[0]
            labs.append(np.zeros(P, np.int32)); offs.append(np.zeros((P, 4), F32)); ctrs.append(np.zeros(P, F32))
            continue
        idx = areas.argmin(axis=0)
        matched = gtl[idx]
        amin = areas[idx, np.arange(areas.shape[1])]
        lab = matched[:, 4].astype(np.int32)
        lab[np.isinf(amin)] = 0
        off = point_encode(all_pts, matched[:, :4])
        lr = off[:, [0, 2]]
        tb = off[:, [1, 3]]
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.sqrt(np.fmax(lr.min(1) / lr.max(1), F32(0)) * np.fmax(tb.min(1) / tb.max(1), F32(0))).astype(F32)
        labs.append(lab); offs.append(off); ctrs.append(c)
    return np.stack(labs), np.stack(offs), np.stack(ctrs)


# --------------------------------------------------------------------------------------------
# OTA target assignment, top-k matcher (models/det/ota.py:76-181, layers/common/matcher.py:123-161)
# --------------------------------------------------------------------------------------------
def _logsumexp(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True, dtype=F32))).squeeze(axis).astype(F32)


def sinkhorn_match(cost, ious, eps=0.1, max_iter=50, topq=20):
    """SinkhornMatcher.__call__ + SinkhornDistance.forward (layers/common/matcher.py:106-121, layers/blocks/sinkhorn_distance.py:22-50)
    in float32.  cost (G+1, P) with the background row last, ious (G, P).  Returns the matched row per point."""
    cost = np.asarray(cost, F32)
    P = cost.shape[1]
    kk = min(topq, P)
    topk = -np.sort(-np.asarray(ious, F32), axis=1, kind="stable")[:, :kk]
    mu = []
    for row in topk:
        s = F32(0)
        for v in row:
            s = F32(s + v)
        mu.append(F32(max(1, int(s))))
    mu = np.asarray(mu + [F32(P) - np.asarray(mu, F32).sum(dtype=F32)], F32)
    nu = np.ones(P, F32)
    u, v = np.ones_like(mu), np.ones_like(nu)
    e = F32(eps)

    def M(u, v):
        return ((-cost + u[:, None] + v[None, :]) / e).astype(F32)
    for _ in range(max_iter):
        v = (v + e * (np.log(nu + F32(1e-8)) - _logsumexp(M(u, v).T, -1))).astype(F32)
        u = (u + e * (np.log(mu + F32(1e-8)) - _logsumexp(M(u, v), -1))).astype(F32)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        pi = np.exp(M(u, v)).astype(F32)
        pi = pi / pi.max(axis=1, keepdims=True)
    return pi.argmax(axis=0), pi


def ota_ground_truth(points_list, strides, logits, pred_ltrb, batched_gt_boxes, num_valid, alpha=0.25, gamma=2.0,
                     reg_weight=1.5, center_radius=2.5, candidate_k=10, matching="topk"):
    """logits (N, P, K), pred_ltrb (N, P, 4) float32 (detached predictions).  Returns labels (N, P) int32 (class, 0 = bg), ltrb
    targets (N, P, 4), IoU targets (N, P), and the cost / IoU matrices per image (for tie analysis in the tests).
    Unpinned orders fixed here: F.topk ties -> lowest point index, F.argmin ties -> lowest gt index."""
    all_pts = np.concatenate(points_list, axis=0).astype(F32)
    P = all_pts.shape[0]
    eps = np.finfo(np.float32).eps
    labs, tgts, ious_out, aux = [], [], [], []
    for n, (boxes_with_labels, nv) in enumerate(zip(batched_gt_boxes, num_valid)):
        gtl = np.asarray(boxes_with_labels, F32)[: int(nv)]
        G = gtl.shape[0]
        lab = np.zeros(P, np.int32); tgt = np.zeros((P, 4), F32); iou_t = np.zeros(P, F32)
        if G == 0:
            labs.append(lab); tgts.append(tgt); ious_out.append(iou_t); aux.append((np.zeros((0, P), F32), np.zeros((0, P), F32)))
            continue
        gt = gtl[:, :4]
        deltas = point_encode(all_pts[None, :, :], gt[:, None, :])                     # (G, P, 4)
        in_boxes = deltas.min(axis=-1) > F32(0.01)
        ctr = ((gt[:, :2] + gt[:, 2:4]) / F32(2)).astype(F32)
        in_ctr = []
        for stride, pts in zip(strides, points_list):
            r = F32(stride * center_radius)
            cb = np.concatenate([np.maximum(ctr - r, gt[:, :2]), np.minimum(ctr + r, gt[:, 2:4])], axis=-1)
            in_ctr.append(point_encode(np.asarray(pts, F32)[None], cb[:, None, :]).min(axis=-1) > 0)
        in_boxes &= np.concatenate(in_ctr, axis=1)
        x = np.asarray(logits[n], F32)                                                 # (P, K)
        K = x.shape[1]
        onehot = np.zeros((G, K), F32)
        onehot[np.arange(G), gtl[:, 4].astype(np.int32) - 1] = 1
        loss_cls = np.stack([sigmoid_focal_loss(x, np.broadcast_to(onehot[g][None], x.shape), alpha, gamma).sum(axis=-1)
                             for g in range(G)]).astype(F32)                          # (G, P)
        ious = ltrb_iou(np.broadcast_to(np.asarray(pred_ltrb[n], F32)[None], deltas.shape), deltas, "iou", eps).astype(F32)
        loss_delta = (-np.log(np.maximum(ious, eps))).astype(F32)
        cost = (loss_cls + F32(reg_weight) * loss_delta + F32(1e6) * (~in_boxes).astype(F32)).astype(F32)
        if matching == "sinkhorn":                                                       # ota.py:153-157
            loss_cls_bg = sigmoid_focal_loss(x, np.zeros_like(x), alpha, gamma).sum(axis=-1).astype(F32)
            cost_bg = np.concatenate([cost, loss_cls_bg[None]], 0)
            ious_m = (ious * in_boxes.astype(F32)).astype(F32)
            mg, pi = sinkhorn_match(cost_bg, ious_m)
            fg = mg != G
            lab[fg] = gtl[mg[fg], 4].astype(np.int32)
            tgt[fg] = deltas[mg[fg], np.nonzero(fg)[0]]
            iou_t[fg] = ious_m[mg[fg], np.nonzero(fg)[0]]
            labs.append(lab); tgts.append(tgt); ious_out.append(iou_t); aux.append((cost_bg, pi))
            continue
        # OTATopkMatcher
        mm = np.zeros((G, P), np.int32)
        kk = min(candidate_k, P)
        topk = -np.sort(-ious, axis=1, kind="stable")[:, :kk]
        dyn = []
        for g in range(G):
            s = F32(0)
            for v in topk[g]:
                s = F32(s + v)
            dyn.append(max(1, int(s)))
        for g in range(G):
            idx = np.argsort(cost[g], kind="stable")[: dyn[g]]
            mm[g, idx] = 1
        multi = mm.sum(0) > 1
        if multi.any():
            am = cost[:, multi].argmin(axis=0)
            mm[:, multi] = 0
            mm[am, np.nonzero(multi)[0]] = 1
        fg = mm.sum(0) > 0
        mg = mm.argmax(axis=0)
        lab[fg] = gtl[mg[fg], 4].astype(np.int32)
        tgt[fg] = deltas[mg[fg], np.nonzero(fg)[0]]
        iou_t[fg] = ious[mg[fg], np.nonzero(fg)[0]]
        labs.append(lab); tgts.append(tgt); ious_out.append(iou_t); aux.append((cost, ious))
    return np.stack(labs), np.stack(tgts), np.stack(ious_out), aux


# --------------------------------------------------------------------------------------------
# ATSS target assignment (models/det/atss.py:17-86)
# --------------------------------------------------------------------------------------------
def atss_ground_truth(points_list, strides, batched_gt_boxes, num_valid, anchor_scale=8, topk=9):
    """Returns labels (N, P) int32, ltrb offsets (N, P, 4) f32, centerness (N, P) f32.
    Unpinned choices: F.topk(distances, descending=False) ties -> lowest index; F.std = population standard deviation
    (MegEngine: sqrt(mean((x - mean)^2))); mean / std are accumulated in float32 in candidate order (level-major, then rank);
    argmax over the gt axis -> lowest gt index; points matched to nothing report gt 0 (argmax of an all -1 column)."""
    all_pts = np.concatenate(points_list, axis=0).astype(F32)
    P = all_pts.shape[0]
    labs, offs, ctrs = [], [], []
    for boxes_with_labels, n in zip(batched_gt_boxes, num_valid):
        gtl = np.asarray(boxes_with_labels, F32)[: int(n)]
        if gtl.shape[0] == 0:
            labs.append(np.zeros(P, np.int32)); offs.append(np.zeros((P, 4), F32)); ctrs.append(np.zeros(P, F32))
            continue
        gt = gtl[:, :4]
        G = gt.shape[0]
        ctr = ((gt[:, :2] + gt[:, 2:4]) / F32(2)).astype(F32)
        ious = np.full((G, P), F32(-1), F32)
        base = 0
        cand_idx, cand_iou = [], []
        for stride, pts in zip(strides, points_list):
            pts = np.asarray(pts, F32)
            hs = F32(stride * anchor_scale / 2)
            anchors = np.concatenate([pts - hs, pts + hs], axis=1).astype(F32)
            iou_l = box_iou(gt, anchors)                                            # (G, P_l)
            d = ctr[:, None, :] - pts[None, :, :]
            dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(F32)).astype(F32)
            k = min(topk, pts.shape[0])
            order = np.argsort(dist, axis=1, kind="stable")[:, :k]                  # ascending, ties -> lowest index
            cand_idx.append(base + order)
            cand_iou.append(np.take_along_axis(iou_l, order, axis=1))
            base += pts.shape[0]
        cand_idx = np.concatenate(cand_idx, axis=1)
        cand_iou = np.concatenate(cand_iou, axis=1).astype(F32)
        for g in range(G):
            s = F32(0)
            for v in cand_iou[g]:
                s = F32(s + v)
            mean = F32(s / F32(cand_iou.shape[1]))
            var = F32(0)
            for v in cand_iou[g]:
                dv = F32(v - mean)
                var = F32(var + F32(dv * dv))
            thr = F32(mean + np.sqrt(F32(var / F32(cand_iou.shape[1]))).astype(F32))
            fg = cand_iou[g] >= thr
            pts_c = all_pts[cand_idx[g]]
            ltrb = np.stack([pts_c[:, 0] - gt[g, 0], pts_c[:, 1] - gt[g, 1], gt[g, 2] - pts_c[:, 0], gt[g, 3] - pts_c[:, 1]], 1)
            inb = ltrb.min(axis=1) > 0
            keep = fg & inb
            ious[g, cand_idx[g][keep]] = cand_iou[g][keep]
        idx = ious.argmax(axis=0)
        amax = ious[idx, np.arange(P)]
        matched = gtl[idx]
        lab = matched[:, 4].astype(np.int32)
        lab[amax == F32(-1)] = 0
        off = point_encode(all_pts, matched[:, :4])
        lr = off[:, [0, 2]]
        tb = off[:, [1, 3]]
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.sqrt(np.fmax(lr.min(1) / lr.max(1), F32(0)) * np.fmax(tb.min(1) / tb.max(1), F32(0))).astype(F32)
        labs.append(lab); offs.append(off); ctrs.append(c)
    return np.stack(labs), np.stack(offs), np.stack(ctrs)


# --------------------------------------------------------------------------------------------
# losses (layers/losses/*.py) -- elementwise values; float64 versions used for gradient checks
# --------------------------------------------------------------------------------------------
def _logsigmoid(x):
    return -np.logaddexp(0, -x)


def binary_cross_entropy(pred, label):
    """cross_entropy.py:7-29 (with_logits=True, logsigmoid form :26)."""
    pred = np.asarray(pred, np.float64)
    label = np.asarray(label, np.float64)
    return -(label * _logsigmoid(pred) + (1 - label) * _logsigmoid(-pred))


def sigmoid_focal_loss(logits, targets, alpha=-1, gamma=0):
    """sigmoid_focal_loss.py:9-36 (float64 evaluation of the same expression)."""
    x = np.asarray(logits, np.float64)
    t = np.asarray(targets, np.float64)
    p = 1.0 / (1.0 + np.exp(-x))
    loss = binary_cross_entropy(x, t)
    if gamma != 0:
        loss = loss * (t * (1 - p) + (1 - t) * p) ** gamma
    if alpha >= 0:
        loss = loss * (t * alpha + (1 - t) * (1 - alpha))
    return loss


def sigmoid_focal_loss_grad(logits, targets, alpha=0.25, gamma=2.0):
    """Analytic d loss / d logit (float64) of the expression above."""
    x = np.asarray(logits, np.float64)
    t = np.asarray(targets, np.float64)
    p = 1.0 / (1.0 + np.exp(-x))
    ce = binary_cross_entropy(x, t)
    pt_ = t * (1 - p) + (1 - t) * p          # "1 - p_t"
    a = t * alpha + (1 - t) * (1 - alpha) if alpha >= 0 else 1.0
    dce = p - t
    dpt = (1 - 2 * t) * p * (1 - p)
    if gamma != 0:
        g = a * (dce * pt_ ** gamma + ce * gamma * pt_ ** (gamma - 1) * dpt)
    else:
        g = a * dce
    return g


def smooth_l1_loss(pred, target, beta=1.0):
    """smooth_l1_loss.py:7-34 (beta < 1e-5 -> pure L1 :28)."""
    x = np.asarray(pred, np.float64) - np.asarray(target, np.float64)
    ax = np.abs(x)
    if beta < 1e-5:
        return ax
    return np.where(ax < beta, 0.5 * x ** 2 / beta, ax - 0.5 * beta)


def ltrb_iou(b1, b2, iou_type="iou", eps=1e-8):
    """get_ltrb_boxes_iou (iou_loss.py:9-56)."""
    b1 = np.asarray(b1, np.float64)
    b2 = np.asarray(b2, np.float64)
    b1 = np.concatenate([-b1[..., :2], b1[..., 2:]], -1)
    b2 = np.concatenate([-b2[..., :2], b2[..., 2:]], -1)
    a1 = np.clip(b1[..., 2] - b1[..., 0], 0, None) * np.clip(b1[..., 3] - b1[..., 1], 0, None)
    a2 = np.clip(b2[..., 2] - b2[..., 0], 0, None) * np.clip(b2[..., 3] - b2[..., 1], 0, None)
    wi = np.clip(np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0]), 0, None)
    hi = np.clip(np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1]), 0, None)
    ai = wi * hi
    au = a1 + a2 - ai
    ious = ai / np.clip(au, eps, None)
    if iou_type == "iou":
        return ious
    gw = np.maximum(b1[..., 2], b2[..., 2]) - np.minimum(b1[..., 0], b2[..., 0])
    gh = np.maximum(b1[..., 3], b2[..., 3]) - np.minimum(b1[..., 1], b2[..., 1])
    ac = gw * gh
    return ious - (ac - au) / np.clip(ac, eps, None)


def iou_loss_ltrb(pred, target, loss_type="giou", eps=1e-8):
    """iou_loss (iou_loss.py:59-105) with box_mode='ltrb'."""
    iou_type = "iou" if loss_type == "linear_iou" else loss_type
    if loss_type == "square_iou":
        iou_type = "iou"
    ious = ltrb_iou(pred, target, iou_type, eps)
    if loss_type == "iou":
        return -np.log(np.clip(ious, eps, None))
    if loss_type == "square_iou":
        return 1 - ious ** 2
    return 1 - ious


# --------------------------------------------------------------------------------------------
# NMS (layers/common/post_processing.py)
# --------------------------------------------------------------------------------------------
def nms(boxes, scores, iou_thresh, max_output=None):
    """Greedy NMS in descending-score order (stable: ties keep the lower index first).
    Suppression rule ``iou > thresh`` (py_cpu_nms keeps ``iou <= thresh`` post_processing.py:130);
    IoU = inter / (a_i + a_j - inter), float32, no +1.  Returns indices in descending-score order."""
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    order = np.argsort(-scores, kind="stable")
    areas = box_area(boxes)
    keep = []
    suppressed = np.zeros(len(boxes), bool)
    for i in order:
        if suppressed[i]:
            continue
        keep.append(int(i))
        if max_output is not None and len(keep) >= max_output:
            break
        xx1 = np.maximum(boxes[i, 0], boxes[:, 0]); yy1 = np.maximum(boxes[i, 1], boxes[:, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[:, 2]); yy2 = np.minimum(boxes[i, 3], boxes[:, 3])
        inter = np.maximum(xx2 - xx1, F32(0)) * np.maximum(yy2 - yy1, F32(0))
        union = (areas[i] + areas) - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / union
        suppressed |= iou > F32(iou_thresh)
    return np.asarray(keep, dtype=np.int32)


def batched_nms(boxes, scores, idxs, iou_thresh, max_output=None):
    """batched_nms (post_processing.py:17-47): class-offset trick ``idxs * (max_coord + 1)`` (:44-46)."""
    boxes = np.asarray(boxes, F32)
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int32)
    idxs = np.asarray(idxs)
    max_c = boxes.max()
    offsets = idxs.astype(F32) * (max_c + F32(1))
    return nms(boxes + offsets[:, None], scores, iou_thresh, max_output)


# --------------------------------------------------------------------------------------------
# pre-processing (layers/common/pre_processing.py)
# --------------------------------------------------------------------------------------------
def get_multiple_size(n, multiple=32):
    return (n + multiple - 1) // multiple * multiple


def get_padded_tensor(t, multiple=32, pad_value=0.0):
    """pre_processing.py:26-49: top-left aligned padding to a multiple of ``multiple``."""
    t = np.asarray(t)
    *size, h, w = t.shape
    out = np.full((*size, get_multiple_size(h, multiple), get_multiple_size(w, multiple)), pad_value, dtype=t.dtype)
    out[..., :h, :w] = t
    return out


def data_to_input(image, mean=None, std=None):
    """pre_processing.py:11-19: pad with 0 FIRST, then normalise => pad region = -mean/std."""
    x = get_padded_tensor(np.asarray(image, F32), 32, 0.0)
    if mean is not None:
        x = x - np.asarray(mean, F32).reshape(1, -1, 1, 1)
    if std is not None:
        x = x / np.asarray(std, F32).reshape(1, -1, 1, 1)
    return x.astype(F32)


# --------------------------------------------------------------------------------------------
# synthetic batch (utils/dummy.py:8-63) -- annotation pattern captured in tests/golden/dummy_loader.npz
# --------------------------------------------------------------------------------------------
def tile_batch(x, batch_size):
    """dummy.py:51-57 repeat/remainder tiling."""
    x = np.asarray(x)
    repeat = batch_size // len(x)
    remain = batch_size % len(x)
    return np.concatenate([np.repeat(x, repeat, axis=0), x[:remain]], axis=0)
