"""Per-layer max |g| of the gradients the fp8 data-gradient launches consume (RetinaNet-R50 / argv[1] = resnet101, 800x1344, batch 4, one step at initialisation):
how far apart are the layers that today share ONE e5m2 scale?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.utils import DummyLoader
B = 4
BACKBONE = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = B; cfg.MODEL.BACKBONE.NAME = BACKBONE; cfg.MODEL.WEIGHT_DTYPE = "fp8_e4m3"; cfg.MODEL.FP8_DGRAD = True; cfg.MODEL.FP8_AMAX_INTERVAL = 1; cfg.MODEL.FP8_AMAX_DELAY = 1
params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
m = RetinaNet(cfg, params=params)
b = next(DummyLoader(B, (800, 1344), seed=0))
batch = {"data": torch.from_numpy(b["data"].astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(), "im_info": torch.from_numpy(b["im_info"]).cuda()}
m(batch); m.backward(); torch.cuda.synchronize()
am = m._amax_host.numpy().copy()
rows = sorted(zip(am, [c.name for c in m._fp8_grad_layers]), reverse=True)
for a, n in rows:
    print(f"{n:50s} {a:.3e}  2^{np.log2(a) if a > 0 else float('-inf'):.1f}")
nz = am[am > 0]
print("max / min over the layers that were probed: 2^%.1f" % np.log2(nz.max() / nz.min()))
