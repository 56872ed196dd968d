"""End-to-end sanity of train -> inference -> COCO box-AP on the HIP path (basedet_amd/evaluators/selfcheck.py holds the code; the
asserted form is tests/test_overfit_ap_gpu.py): a detector is trained on ONE repeated DummyLoader batch and every image of that batch
goes through `model.inference` and `COCOEvaluator` against the batch's own annotations.
usage: python scripts/overfit_ap.py [steps=1500] [batch=16] [backbone=resnet50] [H=800] [W=1344] [lr_per_image=cfg] [warm=cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from basedet_amd.configs import RetinaNetConfig, retinanet_r18_config
from basedet_amd.evaluators.selfcheck import overfit
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.utils import DummyLoader

a = sys.argv[1:]
steps = int(a[0]) if len(a) > 0 else 1500
B = int(a[1]) if len(a) > 1 else 16
backbone = a[2] if len(a) > 2 else "resnet50"
H, W = (int(a[3]), int(a[4])) if len(a) > 4 else (800, 1344)
cfg = retinanet_r18_config() if backbone == "resnet18" else RetinaNetConfig()
cfg.MODEL.BATCHSIZE = B
if len(a) > 5:
    cfg.SOLVER.BASIC_LR = float(a[5])
if len(a) > 6:
    cfg.SOLVER.WARM_ITERS = int(a[6])
params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
model = RetinaNet(cfg, params=params)
if os.environ.get("BD_PAINTED") == "1":          # the asserted test's batch (evaluators/selfcheck.painted_batch): two images
    from basedet_amd.evaluators.selfcheck import painted_batch
    assert B == 2
    hb = painted_batch((H, W))
else:
    hb = next(DummyLoader(B, (H, W), seed=0))
    hb["data"] = (hb["data"] * 255).astype(np.float32)        # pixel range 0..255 (DummyLoader draws [0, 1): next to the dataset mean that is a constant image)
overfit(cfg, model, hb, steps, eval_every=max(steps // 6, 1), log=lambda s: print(s, flush=True))
