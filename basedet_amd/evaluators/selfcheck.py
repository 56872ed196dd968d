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

__all__ = ["batch_annotations", "evaluate_batch", "overfit"]


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
