"""COCO box-AP evaluation for the inference path (`basedet/evaluators/coco_eval.py:72-172`).

The reference's evaluator formats detections as COCO result dicts and hands them to `pycocotools.cocoeval.COCOeval`
(third party, unpinned in requirements.txt, not installed here and not under /root/reference).  `bbox_eval` restates COCOeval's
published bbox protocol in numpy: per (image, category) greedy matching of score-sorted detections at IoU 0.50:0.05:0.95, crowd
ground truth matched any number of times with IoU = inter / det area and then ignored, ground truth outside the area range
ignored (and unmatched detections outside it), at most maxDets detections per image, 101-point interpolated precision
envelope, -1 where a cell has no ground truth.  Host-side numpy; parity with pycocotools is unpinned (no copy of it here): the
tests pin hand-computable cases."""
import json

import numpy as np

__all__ = ["COCOEvaluator", "bbox_eval"]

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}
MAX_DETS = (1, 10, 100)


def _iou_xywh(dt, gt, crowd):
    """dt (D,4), gt (G,4) in xywh; crowd (G,) bool -> (D,G); a crowd column uses the detection's area as the union."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    dx1, dy1, dx2, dy2 = dt[:, 0, None], dt[:, 1, None], dt[:, 0, None] + dt[:, 2, None], dt[:, 1, None] + dt[:, 3, None]
    gx1, gy1, gx2, gy2 = gt[None, :, 0], gt[None, :, 1], gt[None, :, 0] + gt[None, :, 2], gt[None, :, 1] + gt[None, :, 3]
    iw = np.clip(np.minimum(dx2, gx2) - np.maximum(dx1, gx1), 0, None)
    ih = np.clip(np.minimum(dy2, gy2) - np.maximum(dy1, gy1), 0, None)
    inter = iw * ih
    da = (dt[:, 2] * dt[:, 3])[:, None]
    ga = (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(crowd[None, :], da, da + ga - inter)
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def _evaluate_img(dts, gts, rng, max_det):
    """One (image, category) cell -> (scores, dt_matched (T,D) bool, dt_ignore (T,D) bool, number of non-ignored gts)."""
    g_ign = np.array([bool(g.get("iscrowd", 0)) or g["area"] < rng[0] or g["area"] > rng[1] for g in gts], dtype=bool)
    g_order = np.argsort(g_ign, kind="mergesort")                      # non-ignored first
    gts = [gts[i] for i in g_order]; g_ign = g_ign[g_order]
    d_order = np.argsort([-d["score"] for d in dts], kind="mergesort")[:max_det]
    dts = [dts[i] for i in d_order]
    crowd = np.array([bool(g.get("iscrowd", 0)) for g in gts], dtype=bool)
    db = np.array([d["bbox"] for d in dts], dtype=np.float64).reshape(-1, 4)
    gb = np.array([g["bbox"] for g in gts], dtype=np.float64).reshape(-1, 4)
    ious = _iou_xywh(db, gb, crowd)
    T, D, G = len(IOU_THRS), len(dts), len(gts)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = -np.ones((T, D), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(IOU_THRS):
        for di in range(D):
            best, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] >= 0 and not crowd[gi]:
                    continue                                              # taken (a crowd region may match again)
                if m > -1 and not g_ign[m] and g_ign[gi]:
                    break                                                 # a real match exists: do not trade it for an ignored one
                if ious[di, gi] < best:
                    continue
                best, m = ious[di, gi], gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ign[m]
            dtm[ti, di] = m
            gtm[ti, m] = di
    d_area = np.array([d.get("area", d["bbox"][2] * d["bbox"][3]) for d in dts], dtype=np.float64)
    out_rng = (d_area < rng[0]) | (d_area > rng[1])
    dt_ig |= (dtm < 0) & out_rng[None, :]
    return np.array([d["score"] for d in dts], dtype=np.float64), dtm >= 0, dt_ig, int((~g_ign).sum())


def bbox_eval(gt_anns, dt_anns, img_ids=None, cat_ids=None):
    """gt_anns: COCO annotation dicts (image_id, category_id, bbox xywh, area, iscrowd); dt_anns: result dicts (image_id,
    category_id, bbox xywh, score).  Returns {"stats": the 12 COCO numbers, "precision": (T,R,K,A,M), "recall": (T,K,A,M)}."""
    img_ids = sorted(set(img_ids if img_ids is not None else [a["image_id"] for a in gt_anns] + [d["image_id"] for d in dt_anns]))
    cat_ids = sorted(set(cat_ids if cat_ids is not None else [a["category_id"] for a in gt_anns]))
    gts, dts = {}, {}
    for a in gt_anns:
        gts.setdefault((a["image_id"], a["category_id"]), []).append(a)
    for d in dt_anns:
        dts.setdefault((d["image_id"], d["category_id"]), []).append(d)
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(cat_ids), len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for ki, cat in enumerate(cat_ids):
        for ai, rng in enumerate(AREA_RNG.values()):
            cells = [_evaluate_img(dts.get((im, cat), []), gts.get((im, cat), []), rng, MAX_DETS[-1])
                     for im in img_ids if (im, cat) in gts or (im, cat) in dts]
            for mi, md in enumerate(MAX_DETS):
                if not cells:
                    continue
                scores = np.concatenate([c[0][:md] for c in cells])
                order = np.argsort(-scores, kind="mergesort")
                dtm = np.concatenate([c[1][:, :md] for c in cells], axis=1)[:, order]
                dig = np.concatenate([c[2][:, :md] for c in cells], axis=1)[:, order]
                npig = sum(c[3] for c in cells)
                if npig == 0:
                    continue
                tps = np.cumsum(dtm & ~dig, axis=1, dtype=np.float64)
                fps = np.cumsum(~dtm & ~dig, axis=1, dtype=np.float64)
                for ti in range(T):
                    tp, fp = tps[ti], fps[ti]
                    rc = tp / npig
                    pr = tp / (tp + fp + np.spacing(1))
                    recall[ti, ki, ai, mi] = rc[-1] if len(tp) else 0
                    for i in range(len(pr) - 1, 0, -1):                     # precision envelope
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    inds = np.searchsorted(rc, REC_THRS, side="left")
                    q = np.zeros(R)
                    ok = inds < len(pr)
                    q[ok] = pr[inds[ok]]
                    precision[ti, :, ki, ai, mi] = q

    def _mean(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0

    ai_of = {k: i for i, k in enumerate(AREA_RNG)}
    t50, t75 = 0, 5
    stats = [
        _mean(precision[:, :, :, ai_of["all"], 2]), _mean(precision[t50, :, :, ai_of["all"], 2]),
        _mean(precision[t75, :, :, ai_of["all"], 2]), _mean(precision[:, :, :, ai_of["small"], 2]),
        _mean(precision[:, :, :, ai_of["medium"], 2]), _mean(precision[:, :, :, ai_of["large"], 2]),
        _mean(recall[:, :, ai_of["all"], 0]), _mean(recall[:, :, ai_of["all"], 1]), _mean(recall[:, :, ai_of["all"], 2]),
        _mean(recall[:, :, ai_of["small"], 2]), _mean(recall[:, :, ai_of["medium"], 2]), _mean(recall[:, :, ai_of["large"], 2]),
    ]
    return {"stats": stats, "precision": precision, "recall": recall}


STAT_NAMES = ("AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl")


class COCOEvaluator:
    """postprocess -> format -> save_results -> evaluate, as `coco_eval.py:72-172`.  `category_ids`: contiguous label
    (0-based) -> dataset category id (the reference's `classes_originID` lookup, :131-136); default label + 1."""

    def __init__(self, cfg=None, category_ids=None):
        self.cfg = cfg
        self.category_ids = category_ids

    def postprocess(self, model_outputs, image_id):
        """model_outputs: dict with boxes (D,4) xyxy in original-image pixels, box_scores (D,), box_labels (D,) 0-based
        (what `model.inference` returns per image)."""
        boxes = np.asarray(model_outputs["boxes"], dtype=np.float64).reshape(-1, 4)
        if boxes.shape[0] == 0:
            return {"det_res": np.zeros((0, 6)), "image_id": int(image_id)}
        scores = np.asarray(model_outputs["box_scores"], dtype=np.float64).reshape(-1, 1)
        labels = np.asarray(model_outputs["box_labels"], dtype=np.float64).reshape(-1, 1)
        return {"det_res": np.concatenate([boxes, scores, labels], axis=1), "image_id": int(image_id)}

    def format(self, results):
        out = []
        for rec in results:
            boxes = np.array(rec["det_res"], dtype=np.float64)
            if len(boxes) == 0:
                continue
            boxes[:, 2:4] -= boxes[:, 0:2]                                 # xyxy -> xywh (:124)
            for b in boxes:
                lab = int(b[5])
                out.append({"image_id": rec["image_id"], "bbox": b[:4].tolist(), "score": float(b[4]),
                            "category_id": self.category_ids[lab] if self.category_ids is not None else lab + 1})
        return out

    def save_results(self, results, filename):
        with open(filename, "w") as f:
            json.dump(self.format(results), f, indent=4)
        return filename

    def evaluate(self, results, annotations):
        """results: path of a saved result file or the formatted list; annotations: path of / loaded COCO annotation json."""
        if isinstance(results, str):
            with open(results) as f:
                results = json.load(f)
        if isinstance(annotations, str):
            with open(annotations) as f:
                annotations = json.load(f)
        img_ids = [im["id"] for im in annotations["images"]] if "images" in annotations else None
        cat_ids = [c["id"] for c in annotations["categories"]] if "categories" in annotations else None
        res = bbox_eval(annotations["annotations"], results, img_ids, cat_ids)
        return dict(zip(STAT_NAMES, res["stats"]))
