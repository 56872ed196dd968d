"""The RCCL bucket path on ONE GPU: a process group of a single rank sends every gradient bucket through c10d + RCCL on the
communication stream (a SUM over one rank is the identity, the MEAN scale is 1).  The parameters after three training steps must be
bit-identical to a run without the process group: any missing dependency between the producing streams (main, weight-gradient
side stream), the communication stream and the SGD launch would show up as a stale or half-written gradient.  Each run is its own
process (the process group must not leak into the other tests).  The world-size-2 semantics are covered on CPU (test_dist_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
import basedet_amd
if os.environ.get("BD_FORCE_ALLREDUCE") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from basedet_amd.configs import RetinaNetConfig
from basedet_amd.models import RetinaNet, params as P
from basedet_amd.solver import DetSolver
from basedet_amd.utils import DummyLoader
cfg = RetinaNetConfig(); cfg.MODEL.BATCHSIZE = 4
model = RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2))
solver = DetSolver.build(cfg, model)
assert solver.buckets.enabled == (os.environ.get("BD_FORCE_ALLREDUCE") == "1")
solver.optimizer.param_groups[0]["lr"] = 1e-3
b = next(DummyLoader(4, (320, 448), seed=0))
batch = {k: torch.from_numpy(np.asarray(b[k], dtype=np.float32)).cuda() for k in ("data", "gt_boxes", "im_info")}
for _ in range(3):
    out = solver.minimize(model, batch)
torch.cuda.synchronize()
assert np.isfinite(float(out["total_loss"]))
w = model.arena.w.cpu().numpy()
sys.stderr.write("DIGEST " + hashlib.sha256(w.tobytes()).hexdigest() + " " + repr(float(np.abs(w).sum())) + "\n")
sys.stderr.flush()
if dist.is_initialized():
    os._exit(0)          # no process-group teardown: its watchdog threads can hold a finished process for minutes
"""


def _run(force, port):
    env = dict(os.environ, BD_FORCE_ALLREDUCE="1" if force else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT, ROOT, str(port)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stderr.splitlines() if l.startswith("DIGEST ")]
    assert line, r.stderr[-2000:]
    return line[-1].split()[1:]


def test_rccl_bucket_path_is_bit_identical_to_the_local_step():
    plain = _run(False, 29541)
    again = _run(False, 29541)
    assert plain == again, "the step itself is not reproducible"
    forced = _run(True, 29541)
    assert forced == plain, (forced, plain)
