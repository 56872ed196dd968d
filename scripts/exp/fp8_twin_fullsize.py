"""Full-size check of the e5m2 gradient twins: RetinaNet-R50, 800x1344, batch B (argv[1], default 32), fp8 mode, one step from the same
weights with FP8_GRAD_TWINS on and off -- per-tensor cosine of the parameter gradients (twins written by the producing launches vs cast
passes over the bf16 gradients: the same numbers up to double rounding)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.utils import DummyLoader

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = next(DummyLoader(B, (800, 1344), seed=0))
batch = {"data": torch.from_numpy(b["data"].astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(), "im_info": torch.from_numpy(b["im_info"]).cuda()}
grads = {}
for twins in (1, 0):
    cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = B
    cfg.MODEL.WEIGHT_DTYPE = "fp8_e4m3"
    cfg.MODEL.FP8_GRAD_TWINS = bool(twins)
    params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
    m = RetinaNet(cfg, params=params)
    out = m(batch); m.backward(); torch.cuda.synchronize()
    g = m.reference_grads()
    grads[twins] = {k: v.double().cpu() for k, v in g.items()}
    print("twins", twins, "loss", float(out["total_loss"]), flush=True)
    del m, g
    torch.cuda.empty_cache()
a = torch.cat([grads[1][k].reshape(-1) for k in grads[1]]); c = torch.cat([grads[0][k].reshape(-1) for k in grads[1]])
print("global cosine twins vs casts:", float(torch.dot(a, c) / (a.norm() * c.norm())), "norm ratio", float(a.norm() / c.norm()))
worst = []
for k in grads[1]:
    x, y = grads[1][k].reshape(-1), grads[0][k].reshape(-1)
    cs = float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-300))
    worst.append((cs, k, float(x.norm()), float(y.norm())))
worst.sort()
for cs, k, nx, ny in worst[:12]:
    print(f"  {k:50s} cos {cs:.5f}  norm {nx:.3e} vs {ny:.3e}")
