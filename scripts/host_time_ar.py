"""Host enqueue time of a training step with and without the (forced, world-size-1) RCCL gradient all-reduce."""
import os
import sys
import time
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import numpy as np
import torch
import torch.distributed as dist
if os.environ.get("BD_FORCE_ALLREDUCE") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.solver import DetSolver
from basedet_amd.utils import DummyLoader
cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = 16
params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
model = RetinaNet(cfg, params=params)
solver = DetSolver.build(cfg, model)
solver.optimizer.param_groups[0]["lr"] = 1e-5
b = next(DummyLoader(16, (800, 1344), seed=0))
batch = {"data": torch.from_numpy(b["data"].astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(),
         "im_info": torch.from_numpy(b["im_info"]).cuda()}
for _ in range(3):
    solver.minimize(model, batch)
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); solver.minimize(model, batch); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms/step:", [round(h * 1e3, 1) for h in hs], "| loop", round((t1 - t0) * 100, 2), "ms/step; with final sync", round((t2 - t0) * 100, 2))
# where the host time goes inside one step
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); solver.minimize(model, batch); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
