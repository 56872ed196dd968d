import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import test_model_gpu as t
from basedet_amd.models import FasterRCNN
from basedet_amd.solver import DetSolver
for lr in (0.002, 0.0005):
    cfg, params, batch = t._frcnn_setup(2, (128,160), seed=3)
    model = FasterRCNN(cfg, params=params)
    solver = DetSolver.build(cfg, model)
    solver.optimizer.param_groups[0]["lr"] = lr
    vs=[]
    for it in range(10):
        out = solver.minimize(model, batch)
        vs.append([round(float(out[k]),3) for k in ("total_loss","rpn_cls_loss","rpn_reg_loss","rcnn_cls_loss","rcnn_reg_loss")])
    print(lr, vs)
