"""train -> inference -> COCO box-AP on one repeated synthetic batch: the end-to-end consistency check of SURVEY row f2.

No dataset or pretrained weights exist in this environment, so the "box-AP within +-0.1 of the reference" clause cannot be measured.
What CAN be shown without data: a detector trained on ONE batch whose annotations are known reproduces those annotations through the
whole evaluation path -- `model.inference` (scores -> top-k -> decode -> batched NMS -> rescale to the original image size,
basedet/models/det/retinanet.py:172-201, layers/common/post_processing.py:78-103) -> `COCOEvaluator.postprocess / format / evaluate`
(basedet/evaluators/coco_eval.py:72-172).  A wrong class offset, box rescale, NMS rule or evaluator matching shows up as a low AP.

Used by tests/test_overfit_ap_gpu.py (asserted) and scripts/overfit_ap.py (printed)."""
import numpy as np
import torch

from .coco_eval import COCOEvaluator

__all__ = ["batch_annotations", "evaluate_batch", "overfit", "painted_batch"]


def painted_batch(size=(320, 416), seed=0):
    """Two synthetic images in the collator's contract (data (2,3,H,W) float32 0..255, gt_boxes (2,G,5) zero-padded, im_info (2,5)): uniform
    noise with one filled rectangle per annotation, colour = a function of the class.  Unlike `DummyLoader`'s pattern (whose image 0 holds
    two class-52 boxes of IoU 0.63 -- NMS at 0.5 can never return both -- a 38 x 24 px box and 5 : 1 slivers) every box here can be matched
    by an anchor at IoU >= 0.5 and no two boxes overlap by more than 0.2: a detector that has learnt the batch can reach AP ~ 1, so a low
    AP means a broken stage, not an unreachable target.  `im_info` declares an ORIGINAL size different from the padded one, so the
    rescale of post_processing.py:93-101 is part of what is checked."""
    H, W = size
    rng = np.random.default_rng(seed)
    # (x1, y1, x2, y2) as fractions of (W, H), class (1-based)
    spec = [[(0.05, 0.08, 0.30, 0.45, 3), (0.40, 0.10, 0.62, 0.38, 17), (0.70, 0.05, 0.95, 0.50, 41), (0.08, 0.58, 0.40, 0.92, 58),
             (0.50, 0.55, 0.68, 0.90, 66), (0.74, 0.62, 0.93, 0.88, 80)],
            [(0.10, 0.10, 0.45, 0.55, 17), (0.55, 0.12, 0.90, 0.42, 3), (0.15, 0.65, 0.38, 0.93, 25), (0.52, 0.55, 0.88, 0.92, 72)]]
    G = max(len(v) for v in spec)
    data = rng.random((2, 3, H, W)) * 64.0 + 96.0                     # noise around mid-grey
    gt = np.zeros((2, G, 5), np.float32)
    for i, boxes in enumerate(spec):
        for k, (fx1, fy1, fx2, fy2, c) in enumerate(boxes):
            x1, y1, x2, y2 = round(fx1 * W), round(fy1 * H), round(fx2 * W), round(fy2 * H)
            col = np.array([(37 * c) % 256, (91 * c + 60) % 256, (53 * c + 130) % 256], np.float64)
            data[i, :, y1:y2, x1:x2] = col[:, None, None] + rng.random((3, y2 - y1, x2 - x1)) * 16.0
            gt[i, k] = (x1, y1, x2, y2, c)
    info = np.array([[H, W, 0.75 * H, 0.75 * W, len(spec[0])], [H, W, 1.25 * H, 1.25 * W, len(spec[1])]], np.float32)
    return {"data": data.astype(np.float32), "gt_boxes": gt, "im_info": info}


def batch_annotations(host_batch):
    """COCO-style ground truth of a collated batch, in ORIGINAL-image coordinates (post_processing.py:93-101 rescales detections by
    orig / padded size, im_info = [H, W, origH, origW, num_gt])."""
    anns, aid = [], 0
    B = host_batch["im_info"].shape[0]
    for i in range(B):
        info = host_batch["im_info"][i]
        sy, sx = info[2] / info[0], info[3] / info[1]
        for g in host_batch["gt_boxes"][i][: int(info[4])]:
            x1, y1, x2, y2 = g[0] * sx, g[1] * sy, g[2] * sx, g[3] * sy
            aid += 1
            anns.append({"id": aid, "image_id": i + 1, "category_id": int(g[4]), "bbox": [float(x1), float(y1), float(x2 - x1), float(y2 - y1)],
                         "area": float((x2 - x1) * (y2 - y1)), "iscrowd": 0})
    return {"annotations": anns, "images": [{"id": i + 1} for i in range(B)]}


def evaluate_batch(model, cfg, host_batch, dev_batch):
    """Every image of the batch through model.inference and the evaluator; returns (stats dict, number of detections)."""
    was_training = model.training
    model.eval()
    ev = COCOEvaluator(cfg)
    results = []
    B = host_batch["im_info"].shape[0]
    for i in range(B):
        out = model.inference({"data": dev_batch["data"][i:i + 1], "im_info": dev_batch["im_info"][i:i + 1]})
        n = len(out.box_scores) if hasattr(out, "box_scores") else 0
        results.append(ev.postprocess({"boxes": out.boxes.cpu().numpy() if n else np.zeros((0, 4)),
                                       "box_scores": out.box_scores.cpu().numpy() if n else np.zeros((0,)),
                                       "box_labels": out.box_labels.cpu().numpy() if n else np.zeros((0,))}, image_id=i + 1))
    stats = ev.evaluate(ev.format(results), batch_annotations(host_batch))
    if was_training:
        model.train()
    return stats, sum(len(r["det_res"]) for r in results)


def overfit(cfg, model, host_batch, steps, eval_every=0, log=None):
    """`steps` solver steps on the one batch (LR schedule from cfg), evaluating every `eval_every` steps and at the end.
    Returns [(step, loss, stats, n_det)]."""
    from ..solver import DetSolver, WarmupMultiStepLR
    solver = DetSolver.build(cfg, model)
    sched = WarmupMultiStepLR(solver.optimizer, cfg, 1)
    dev = {"data": torch.from_numpy(np.ascontiguousarray(host_batch["data"], dtype=np.float32)).cuda(),
           "gt_boxes": torch.from_numpy(host_batch["gt_boxes"]).cuda(), "im_info": torch.from_numpy(host_batch["im_info"]).cuda()}
    hist = []
    loss = float("nan")
    for it in range(steps):
        sched.step(it)
        out = solver.minimize(model, dev)
        last = it + 1 == steps
        if last or (eval_every and (it + 1) % eval_every == 0):
            loss = float(out["total_loss"])
            stats, ndet = evaluate_batch(model, cfg, host_batch, dev)
            hist.append((it + 1, loss, stats, ndet))
            if log:
                log(f"step {it + 1}: loss {loss:.4f} detections {ndet} " + " ".join(f"{k} {stats[k]:.3f}" for k in ("AP", "AP50", "AP75", "AR100")))
    return hist
