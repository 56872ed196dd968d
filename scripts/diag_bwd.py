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
orc = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True)
# monkeypatch to retain grads
keep = {}
orig_fpn = orc.fpn
def fpn(feats):
    for k, v in feats.items():
        if v.requires_grad: v.retain_grad(); keep[k] = v
    out = orig_fpn(feats)
    for i, o in enumerate(out):
        o.retain_grad(); keep[f"P{i+3}"] = o
    return out
orc.fpn = fpn
orig_head = orc.retina_head
def head(feats):
    lg, of = orig_head(feats)
    for i, (a, b) in enumerate(zip(lg, of)):
        a.retain_grad(); b.retain_grad(); keep[f"logits{i+3}"] = a; keep[f"offs{i+3}"] = b
    return lg, of
orc.retina_head = head
ref_losses, aux = orc.retinanet_losses(batch)
ref_losses["total_loss"].backward()
losses = model(batch)
model.backward()
torch.cuda.synchronize()
pl = model._cur
def lvl(buf, i, c):
    v = buf.float().cpu().view(N, pl.pyr.pix_per_img, -1)
    h, w = pl.sizes[i]
    return v[:, pl.pyr.off[i]: pl.pyr.off[i] + h * w, :c].reshape(N, h, w, c).permute(0, 3, 1, 2)
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
for i in range(5):
    print(f"level {i+3}: dlogits rel={rel(lvl(pl.d_logits, i, 720), keep[f'logits{i+3}'].grad):.4f} doffs rel={rel(lvl(pl.d_offsets, i, 36), keep[f'offs{i+3}'].grad):.4f} g_P rel={rel(lvl(pl.g_P, i, 256), keep[f'P{i+3}'].grad):.4f} |gP|={float(keep[f'P{i+3}'].grad.norm()):.3e}")
for s in (3, 4, 5):
    bi = pl.res[s]
    b = pl.blk[bi]
    g = b.g_out.float().cpu().view(N, b.gout.H[0], b.gout.W[0], -1).permute(0, 3, 1, 2)
    r = keep[f"res{s}"]
    ref = r.grad * (r > 0)
    print(f"res{s}: masked grad rel={rel(g, ref):.4f}")
