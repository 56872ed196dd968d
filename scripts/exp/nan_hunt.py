"""Where does the overfit run (scripts/overfit_ap.py at a batch learning rate >= 0.005) leave the finite range?  Trains RetinaNet-R18 on the
repeated 2 x 320 x 416 batch and, every step, checks losses, logits, offsets, gradient arena and weights for non-finite values and reports
the first offender with magnitudes of the step before."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd.configs import retinanet_r18_config
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.solver import DetSolver, WarmupMultiStepLR
from basedet_amd.utils import DummyLoader

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0025
cfg = retinanet_r18_config(); cfg.MODEL.BATCHSIZE = 2
cfg.SOLVER.BASIC_LR = lr; cfg.SOLVER.WARM_ITERS = 100
model = RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2))
solver = DetSolver.build(cfg, model)
sched = WarmupMultiStepLR(solver.optimizer, cfg, 1)
hb = next(DummyLoader(2, (320, 416), seed=0))
batch = {"data": torch.from_numpy((hb["data"] * 255).astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(hb["gt_boxes"]).cuda(),
         "im_info": torch.from_numpy(hb["im_info"]).cuda()}
prev = None
for it in range(3000):
    sched.step(it)
    out = solver.minimize(model, batch)
    pl = model._cur
    st = dict(step=it, loss={k: float(v) for k, v in out.items()}, logit_absmax=float(pl.logits.float().abs().max()),
              offset_absmax=float(pl.offsets.float().abs().max()), grad_absmax=float(model.arena.g.abs().max()),
              w_absmax=float(model.arena.w.abs().max()), num_fg=int(pl.num_fg.item()))
    bad = [k for k, v in st.items() if k != "loss" and isinstance(v, float) and not np.isfinite(v)] + [k for k, v in st["loss"].items() if not np.isfinite(v)]
    if bad or it % 250 == 0:
        print(st, flush=True)
    if bad:
        print("first non-finite:", bad, "\nprevious step:", prev)
        # which parameter tensors carry the non-finite gradient
        for n, g in model.reference_grads().items():
            if not bool(torch.isfinite(g).all()):
                print("  non-finite gradient in", n, "finite absmax", float(g[torch.isfinite(g)].abs().max()) if bool(torch.isfinite(g).any()) else None)
        break
    prev = st
