import sys, numpy as np, torch
import torch.nn.functional as TF
sys.path.insert(0, '.')
from tests.test_model_gpu import _setup
from basedet_amd.models import RetinaNet, params as P
from basedet_amd import ops
from oracle.model import Oracle
backbone, N, size = "resnet50", 3, (96, 128)
cfg, params, batch = _setup(backbone, N, size)
model = RetinaNet(cfg, params=params)
names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
orc = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True)
keep = {}
def head(feats):
    logits, offsets = [], []
    for li, f in enumerate(feats):
        f.retain_grad(); keep[f"P{li}"] = f
        c = f; b = f
        for i in range(4):
            c = orc._q(TF.relu(orc._conv(c, f"head.cls_subnet.{2*i}", 1, 1))); c.retain_grad(); keep[f"c{i}_{li}"] = c
            b = orc._q(TF.relu(orc._conv(b, f"head.bbox_subnet.{2*i}", 1, 1))); b.retain_grad(); keep[f"b{i}_{li}"] = b
        lg = orc._q(orc._conv(c, "head.cls_score", 1, 1)); lg.retain_grad(); keep[f"lg_{li}"] = lg
        logits.append(lg)
        offsets.append(orc._q(orc._conv(b, "head.bbox_pred", 1, 1)))
    return logits, offsets
orc.retina_head = head
ref_losses, aux = orc.retinanet_losses(batch)
ref_losses["total_loss"].backward()
losses = model(batch)
pl = model._cur
pyr = pl.pyr
def lvl(buf, i):
    c = buf.shape[1]
    v = buf.float().cpu().view(N, pyr.pix_per_img, c)
    h, w = pl.sizes[i]
    return v[:, pyr.off[i]: pyr.off[i] + h * w].reshape(N, h, w, c).permute(0, 3, 1, 2)
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
def report(tag, buf, key, masked):
    out = []
    for li in range(5):
        r = keep[f"{key}_{li}"] if key != "P" else keep[f"P{li}"]
        ref = r.grad * (r > 0) if masked else r.grad
        out.append(f"{rel(lvl(buf, li), ref):.4f}")
    print(tag, " ".join(out))
# forward check of tower acts
for i in range(4):
    print("fwd c%d" % i, " ".join(f"{rel(lvl(pl.cls_act[i], li), keep[f'c{i}_{li}']):.4f}" for li in range(5)))
gA, gB = pl.g_tower
model.cls_score.dgrad(pl.d_logits, pyr, pyr, gA, mask=pl.cls_act[3])
report("g_c3(masked)", gA, "c3", True)
g = gA
for i in (3, 2, 1):
    nxt = gB if g is gA else gA
    model.cls_tower[i].dgrad(g, pyr, pyr, nxt, mask=pl.cls_act[i - 1])
    report(f"g_c{i-1}(masked)", nxt, f"c{i-1}", True)
    g = nxt
model.cls_tower[0].dgrad(g, pyr, pyr, pl.g_P, first=True)
print("cls->P only (no ref)")
