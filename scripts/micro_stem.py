"""Timing of the stem convolution + max-pool at the bench shape (16 x 800 x 1344)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basedet_amd import ops
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.configs import RetinaNetConfig
import numpy as np
cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = 16
model = RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2))
from basedet_amd.utils import DummyLoader
b = next(DummyLoader(16, (800, 1344), seed=0))
batch = {k: torch.from_numpy(np.asarray(b[k], dtype=np.float32)).cuda() for k in ("data", "gt_boxes", "im_info")}
from torch.profiler import profile, ProfilerActivity
for _ in range(2): model(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): model(batch)
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "stem_conv" in e.key or "maxpool" in e.key or "pad_normalize" in e.key:
        print(e.key[:50], round(e.device_time_total / e.count, 1), "us x", e.count)
