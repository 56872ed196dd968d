"""Diagnostic: per-tensor gradient cosine of the fp8-forward R18 step against the bf16 step, under kernel-selection knobs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import _lib
from basedet_amd.models import RetinaNet, params as P
from tests.test_model_gpu import _setup

def run(knob):
    _lib.load().bd_conv_set_patch3x3(knob)
    cfg, params, batch = _setup("resnet18", 2, (128, 160))
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    m16 = RetinaNet(cfg, params=params); m16(batch); m16.backward(); g16 = m16.reference_grads()
    cfg.MODEL.WEIGHT_DTYPE = "fp8_e4m3"
    for kv in sys.argv[1:]:
        k, v = kv.split("="); setattr(cfg.MODEL, k, type(getattr(cfg.MODEL, k, 0))(int(v)))
    m8 = RetinaNet(cfg, params=params); m8(batch); m8.backward(); torch.cuda.synchronize(); g8 = m8.reference_grads()
    a = torch.cat([g8[n].double().reshape(-1) for n in names]); b = torch.cat([g16[n].double().reshape(-1) for n in names])
    print("knob", knob, "cos", float(torch.dot(a, b) / (a.norm() * b.norm())))
    for n in names:
        x, y = g8[n].double().reshape(-1), g16[n].double().reshape(-1)
        c = float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-30))
        if c < 0.97: print("   ", n, tuple(g8[n].shape), "cos %.4f" % c, "norm %.3e vs %.3e" % (float(x.norm()), float(y.norm())), "fp8" if m8.convs.get(n.rsplit(".", 1)[0]) is not None and m8.convs[n.rsplit(".", 1)[0]].fp8 else "")

for k in (1, 1 | 64, 0):
    run(k)
