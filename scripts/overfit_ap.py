"""End-to-end sanity of train -> inference -> COCO box-AP on the HIP path: RetinaNet-R50-FPN is trained on ONE repeated DummyLoader batch
(16 x 800 x 1344, the benchmark's synthetic boxes; random-init weights, no dataset in this container), then every image of that batch goes
through `model.inference` (scores -> top-k -> decode -> batched NMS -> rescale, basedet/models/det/retinanet.py:172-201) and the
detections are scored against the batch's own annotations by `COCOEvaluator` (basedet/evaluators/coco_eval.py:72-172 restated).  The AP
of an overfitted batch says nothing about COCO accuracy; it shows that every stage of the path produces consistent boxes, classes and
coordinates (rescaling to the original image size included), and it is the one AP number this build can produce without data.
usage: python scripts/overfit_ap.py [steps=1500] [batch=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.evaluators import COCOEvaluator
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.solver import DetSolver, WarmupMultiStepLR
from basedet_amd.utils import DummyLoader

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = B
params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
model = RetinaNet(cfg, params=params)
solver = DetSolver.build(cfg, model)
sched = WarmupMultiStepLR(solver.optimizer, cfg, 1)
hb = next(DummyLoader(B, (800, 1344), seed=0))
batch = {"data": torch.from_numpy(hb["data"].astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(hb["gt_boxes"]).cuda(),
         "im_info": torch.from_numpy(hb["im_info"]).cuda()}


def evaluate(tag):
    model.eval()
    ev = COCOEvaluator(cfg)
    results, anns = [], []
    aid = 0
    for i in range(B):
        info = hb["im_info"][i]
        out = model.inference({"data": batch["data"][i:i + 1], "im_info": batch["im_info"][i:i + 1]})
        n = len(out.box_scores) if hasattr(out, "box_scores") else 0
        results.append(ev.postprocess({"boxes": out.boxes.cpu().numpy() if n else np.zeros((0, 4)),
                                       "box_scores": out.box_scores.cpu().numpy() if n else np.zeros((0,)),
                                       "box_labels": out.box_labels.cpu().numpy() if n else np.zeros((0,))}, image_id=i + 1))
        sy, sx = info[2] / info[0], info[3] / info[1]                       # post_processing.py:93-101 rescales to the original size
        for g in hb["gt_boxes"][i][: int(info[4])]:
            x1, y1, x2, y2 = g[0] * sx, g[1] * sy, g[2] * sx, g[3] * sy
            aid += 1
            anns.append({"id": aid, "image_id": i + 1, "category_id": int(g[4]), "bbox": [float(x1), float(y1), float(x2 - x1), float(y2 - y1)],
                         "area": float((x2 - x1) * (y2 - y1)), "iscrowd": 0})
    st = ev.evaluate(ev.format(results), {"annotations": anns, "images": [{"id": i + 1} for i in range(B)]})
    ndet = sum(len(r["det_res"]) for r in results)
    print(f"{tag}: detections {ndet}, ground truths {len(anns)}, " + " ".join(f"{k} {v:.3f}" for k, v in st.items() if k in ("AP", "AP50", "AP75", "AR100")), flush=True)
    model.train()
    return st


evaluate("step 0 (random init)")
for it in range(steps):
    sched.step(it)
    out = solver.minimize(model, batch)
    if (it + 1) % 250 == 0 or it + 1 == steps:
        print(f"step {it + 1}: loss {float(out['total_loss']):.4f}", flush=True)
        if (it + 1) % 500 == 0 or it + 1 == steps:
            evaluate(f"step {it + 1}")
