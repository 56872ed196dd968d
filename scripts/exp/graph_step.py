"""Can the training step be captured into a HIP graph (torch.cuda.CUDAGraph around solver.minimize) and replayed?  Host enqueue time per
step, eager step time, replay step time, and the loss / a weight after the same number of steps both ways."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.solver import DetSolver
from basedet_amd.utils import DummyLoader

def build():
    cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = 16
    params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
    model = RetinaNet(cfg, params=params)
    solver = DetSolver.build(cfg, model)
    solver.optimizer.param_groups[0]["lr"] = 1e-5
    return model, solver

b = next(DummyLoader(16, (800, 1344), seed=0))
batch = {"data": torch.from_numpy(b["data"].astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(), "im_info": torch.from_numpy(b["im_info"]).cuda()}

model, solver = build()
for _ in range(5): out = solver.minimize(model, batch)
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); out = solver.minimize(model, batch); hs.append(time.perf_counter() - a)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("eager: host enqueue ms/step median %.2f, step %.2f ms" % (np.median(hs) * 1e3, (t1 - t0) / 20 * 1e3), "loss", float(out["total_loss"]), flush=True)

model2, solver2 = build()
model2.async_wgrad = os.environ.get('ASYNC', '0') == '1'
print('async_wgrad', model2.async_wgrad, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): out2 = solver2.minimize(model2, batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        out_g = solver2.minimize(model2, batch)
    torch.cuda.synchronize()
    print("captured", flush=True)
    t0 = time.perf_counter()
    for _ in range(19): g.replay()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("graph replay: step %.2f ms" % ((t1 - t0) / 19 * 1e3), "loss", float(out_g["total_loss"]))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:600])
