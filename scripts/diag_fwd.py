import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import _setup
from basedet_amd.models import RetinaNet, params as P
from oracle.model import Oracle, _t
from oracle import box_ops
backbone, N, size = "resnet50", 3, (96, 128)
cfg, params, batch = _setup(backbone, N, size)
model = RetinaNet(cfg, params=params)
names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
a = orc.arch
image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
with torch.no_grad():
    feats = orc.backbone(image)
    pyr = orc.fpn(feats)
losses = model(batch)
pl = model._cur
def cmp(name, got_pm, geo, ref):
    n, c, h, w = ref.shape
    g = got_pm.float().cpu().view(n, h, w, c).permute(0, 3, 1, 2)
    print(f"{name:10s} rel={float((g-ref).norm()/ref.norm()):.4f}  |ref|={float(ref.norm()):.3e} max={float(ref.abs().max()):.3e}")
cmp("stem", pl.pool_out, None, feats["stem"])
for s in (2, 3, 4, 5):
    bi = [i for i, b in enumerate(model.blocks) if b["layer"] == s - 1][-1]
    cmp(f"res{s}", pl.blk[bi].out, None, feats[f"res{s}"])
# per block in layer1
x = feats["stem"]
Pv = pl.P.float().cpu().view(N, pl.pyr.pix_per_img, -1)
for i, ref in enumerate(pyr):
    n, c, h, w = ref.shape
    g = Pv[:, pl.pyr.off[i]: pl.pyr.off[i] + h * w].reshape(n, h, w, c).permute(0, 3, 1, 2)
    print(f"P{i+3} rel={float((g-ref).norm()/ref.norm()):.4f}")
